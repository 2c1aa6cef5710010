#!/usr/bin/env python3
"""Offline costing of TRI hand-out schemes from the per-step run lengths the wave emulator logs (tests/emu, -DMGPU_EMU_STATS build;
scratch/emu/tri_steps.bin: per step cT, the open leaves' run lengths, 0).  For every scheme: trips per step, lane occupancy over the
trips, and the instruction estimate trips * per_trip + per_step.  NOTE: a model of a step taken alone -- a scheme that finishes more of
a run per step also changes which steps come next; the numbers rank schemes, they are not frame times.
usage: python tools/model_tri_handout.py [scratch/emu/tri_steps.bin]"""
import sys
import numpy as np
raw = np.fromfile(sys.argv[1] if len(sys.argv) > 1 else "scratch/emu/tri_steps.bin", np.uint8)
steps, i = [], 0
while i < len(raw):
    c = int(raw[i]); steps.append(raw[i + 1:i + 1 + c].astype(int)); i += c + 2
print("%d TRI steps, %.1f open leaves and %.1f tests due per step" % (len(steps), np.mean([len(s) for s in steps]), np.mean([s.sum() for s in steps])))
CAP = 8


def pow2ceil(x):
    return 1 << max(0, int(np.ceil(np.log2(max(1, x)))))


def built(s):  # 2^sh lanes per leaf by the number of open leaves; at most CAP trips
    c = len(s)
    m = 4 if c <= 16 else (2 if c <= 32 else 1)
    trips = min(CAP, int(np.ceil(s.max() / m)))
    done = np.minimum(s, trips * m).sum()
    return trips, done


def flat(s, cap_trips=CAP):  # (leaf, triangle) pairs dealt one by one
    t = min(cap_trips, int(np.ceil(s.sum() / 64)))
    return t, min(s.sum(), t * 64)


def classes(s):  # power-of-two groups per leaf sized for the smallest trip count whose groups fit 64 lanes
    for t in (1, 2, 3, 4, 5, 6, 8):
        m = np.array([min(16, pow2ceil(np.ceil(x / t))) for x in s])
        if m.sum() <= 64:
            trips = int(np.ceil((s / m).max()))
            return min(trips, CAP), np.minimum(s, m * min(trips, CAP)).sum()
    m = np.ones(len(s), int)
    trips = min(CAP, s.max())
    return trips, np.minimum(s, trips).sum()


for name, fn, per_trip, per_step in (("as built (2 / 4 lanes per leaf)", built, 50, 60), ("flat, segmented merge of 6 steps", flat, 50 + 18 + 14 + 110, 40),
                                     ("flat, merge of 4 steps (runs <= 16)", flat, 50 + 18 + 14 + 75, 40), ("power-of-two groups by run length", classes, 50, 60 + 80)):
    r = [fn(s) for s in steps]
    trips = np.array([a for a, _ in r]); done = np.array([b for _, b in r])
    print("%-40s trips/step %.2f  tests/step %.1f  lane occupancy %.3f  instructions/test %.2f" % (name, trips.mean(), done.mean(), done.sum() / (64.0 * trips.sum()),
          (trips * per_trip + per_step).sum() / done.sum()))
