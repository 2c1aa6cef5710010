#!/usr/bin/env python3
"""Authors the two MagicaVoxel fixtures of this directory (input DATA for the .vox reader tests; nothing here comes from
the reference).  Layout as the reference's reader consumes it (importers/magicavoxel_loader.cc:60-113): "VOX ", int32
version, then chunks {id[4], int32 chunkSize, int32 childChunkSize, content}.

  tiny.vox       no RGBA chunk (the reader then uses MagicaVoxel's default palette), a MAIN parent chunk, an unknown chunk
                 that must be skipped, 7 voxels incl. colour index 0 (-> material id -1) and 255
  tiny_rgba.vox  an RGBA palette chunk, 5 voxels, a SIZE chunk with trailing bytes
"""
import os
import struct

HERE = os.path.dirname(os.path.abspath(__file__))


def chunk(cid, content=b"", children=b""):
    return cid + struct.pack("<ii", len(content), len(children)) + content + children


def xyzi(voxels):
    return chunk(b"XYZI", struct.pack("<i", len(voxels)) + b"".join(struct.pack("<4B", *v) for v in voxels))


def write(name, body):
    with open(os.path.join(HERE, name), "wb") as f:
        f.write(b"VOX " + struct.pack("<i", 150) + body)


# a small L-shaped staircase with a floating voxel
v1 = [(0, 0, 0, 1), (1, 0, 0, 2), (2, 0, 0, 37), (2, 1, 0, 216), (2, 1, 1, 255), (0, 3, 2, 0), (3, 3, 3, 248)]
write("tiny.vox", chunk(b"MAIN", b"", chunk(b"SIZE", struct.pack("<3i", 4, 4, 4)) + chunk(b"NOTE", b"skip me!") + xyzi(v1)))

pal = b"".join(struct.pack("<4B", (37 * i + 11) & 255, (91 * i + 5) & 255, (13 * i + 200) & 255, 255) for i in range(256))
v2 = [(1, 1, 1, 1), (2, 1, 1, 9), (1, 2, 1, 100), (1, 1, 2, 200), (5, 4, 3, 256 - 1)]
write("tiny_rgba.vox", chunk(b"SIZE", struct.pack("<3i", 6, 5, 4) + b"\0\0\0\0") + xyzi(v2) + chunk(b"RGBA", pal))
print("wrote tiny.vox, tiny_rgba.vox")
