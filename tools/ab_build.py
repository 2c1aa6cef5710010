#!/usr/bin/env python3
"""A/B builds of the library with extra -D flags, in parallel.  usage: python tools/ab_build.py name1:"-DX=1 -DY=2" name2:"..."
Outputs mallie_amd/ab/<name>.so (git-ignored; travels to the GPU box); load with MALLIE_MGPU_LIB."""
import os, sys, concurrent.futures as cf
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mallie_amd import build as b
outdir = os.path.join(os.path.dirname(b.LIB), "ab")
os.makedirs(outdir, exist_ok=True)
def one(arg):
    name, flags = arg.split(":", 1)
    out = os.path.join(outdir, name + ".so")
    b.build_variant(out, flags.split())
    return out
with cf.ThreadPoolExecutor(4) as ex:
    for o in ex.map(one, sys.argv[1:]):
        print(o)
