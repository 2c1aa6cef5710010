"""The N>1 path on CPU: world_size-2/3 gloo processes exercise mallie_amd.frame.FrameRenderer's strip partition, the
single gather to rank 0 and the re-interleaving, with the oracle standing in for the device renderer of each rank
(there is no GPU here).  The assembled frame must equal a single-process full-frame render bit for bit, which is the
property that makes the image independent of the GPU count (per-(pixel,pass) seeding)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O
from mallie_amd.frame import FrameRenderer, strip_rows

W, H, MPL, PASSES, SEED = 72, 53, 5, 3, 11   # H deliberately not a multiple of the 8-row strip


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_local(osc, frame, plane):
    def render_local(rows, out, pass_base):
        img = np.zeros((H, W, 3), "<f4")
        # strips are contiguous runs of rows: render each run as a window
        runs = np.split(rows, np.where(np.diff(rows) != 1)[0] + 1)
        for run in runs:
            part, _, _, _ = osc.render(frame, W, H, MPL, PASSES, plane, O.RNG_HASH, seed=SEED, pass_base=pass_base,
                                       window=(0, int(run[0]), W, int(run[-1]) + 1), nthreads=2)
            img[run] = part[run]
        out[: len(rows)] = torch.from_numpy(img[rows])
    return render_local


def _worker(rank, world, port, result_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        osc = O.scene_from_golden("cornell_obj")
        frame = O.camera_frame((0, 0, 20), (0, 0, 0), width=W, height=H)
        plane = osc.plane()
        fr = FrameRenderer(None, frame, W, H, MPL, PASSES, plane, SEED, rank, world, torch.device("cpu"),
                           render_local=_oracle_local(osc, frame, plane))
        out = fr.render(pass_base=2)
        # elapsed-time reduction as bench.py does it: MAX over ranks
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert t.item() == world
        if rank == 0:
            np.save(result_path, out.numpy())
        else:
            assert out is None
        # display path: per-rank tonemap (the oracle's mo_tonemap standing in for k_tonemap), 8-bit gather
        def tonemap_local(strips, passes, mode, out_u8):
            cnt = np.full(strips.shape[:2], passes, "<i4")
            out_u8.copy_(torch.from_numpy(O.tonemap(strips.numpy(), cnt, mode).reshape(out_u8.shape)))
        fr.tonemap_local = tonemap_local
        for mode in (0, 1):
            ldr = fr.render_ldr(mode, pass_base=2)
            if rank == 0:
                np.save(result_path + ".ldr%d.npy" % mode, ldr.numpy())
            else:
                assert ldr is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _worker_inflight(rank, world, port, result_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        osc = O.scene_from_golden("cornell_obj")
        frame = O.camera_frame((0, 0, 20), (0, 0, 0), width=W, height=H)
        plane = osc.plane()
        fr = FrameRenderer(None, frame, W, H, MPL, PASSES, plane, SEED, rank, world, torch.device("cpu"),
                           render_local=_oracle_local(osc, frame, plane), frames_in_flight=2)
        outs = [fr.render(pass_base=b) for b in (0, 3, 6)]  # three frames over two rotating buffer sets
        fr.wait()
        if rank == 0:
            assert outs[0] is outs[2] and outs[0] is not outs[1]  # frame 2 reuses (and overwrote) frame 0's buffers
            np.save(result_path, np.stack([outs[1].numpy(), outs[2].numpy()]))
        else:
            assert all(o is None for o in outs)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_frames_in_flight_rotate_buffers(tmp_path):
    """bench.py's N>1 configuration: consecutive frames alternate between two buffer sets (on a GPU: two streams), every
    rank issues its gathers in frame order.  Frames 1 and 2 (pass_base 3 and 6) must equal single-process renders."""
    path = str(tmp_path / "frames.npy")
    mp.spawn(_worker_inflight, args=(2, _free_port(), path), nprocs=2, join=True)
    got = np.load(path)
    osc = O.scene_from_golden("cornell_obj")
    frame = O.camera_frame((0, 0, 20), (0, 0, 0), width=W, height=H)
    for k, base in enumerate((3, 6)):
        ref, _, _, _ = osc.render(frame, W, H, MPL, PASSES, osc.plane(), O.RNG_HASH, seed=SEED, pass_base=base)
        assert got[k].tobytes() == ref.tobytes()


@pytest.mark.parametrize("world", [2, 3])
def test_strip_partition_gather_reassemble(world, tmp_path):
    path = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), path), nprocs=world, join=True)
    got = np.load(path)
    osc = O.scene_from_golden("cornell_obj")
    frame = O.camera_frame((0, 0, 20), (0, 0, 0), width=W, height=H)
    ref, _, _, _ = osc.render(frame, W, H, MPL, PASSES, osc.plane(), O.RNG_HASH, seed=SEED, pass_base=2)
    assert got.tobytes() == ref.tobytes()
    # display frames gathered as 8-bit strips == the driver's transform of the whole float frame
    cnt = np.full((H, W), PASSES, "<i4")
    for mode, ch in ((0, 3), (1, 4)):
        ldr = np.load(path + ".ldr%d.npy" % mode)
        assert ldr.shape == (H, W, ch) and ldr.dtype == np.uint8
        assert ldr.tobytes() == O.tonemap(ref, cnt, mode).tobytes()


def test_strip_rows_cover_frame_exactly_once():
    for Hh, world, sh in [(1080, 8, 8), (1080, 4, 8), (53, 3, 8), (7, 2, 8), (2160, 8, 16), (100, 1, 8)]:
        seen = np.concatenate([strip_rows(Hh, world, r, sh) for r in range(world)])
        assert sorted(seen.tolist()) == list(range(Hh))
        # local row j of rank r maps to y_first + (j // sh) * period + j % sh, as mgpu_render_strips_device expects
        for r in range(world):
            rows = strip_rows(Hh, world, r, sh)
            j = np.arange(len(rows))
            assert np.array_equal(rows, r * sh + (j // sh) * (sh * world) + j % sh)
