"""Multi-GPU frame rendering: the image is cut into interleaved row strips, one process per GPU renders its strips with
the scene replicated in its HBM, and ONE gather over RCCL/xGMI brings the float RGB strips to rank 0.

torch is used here for what it is good at -- device buffers, streams, torch.distributed -- while all rendering goes
through the C ABI (mallie_amd.mgpu).  Pixels are seeded per (pixel, pass), so the frame is independent of the GPU count.
"""
import numpy as np
import torch

from . import mgpu

STRIP_H = 8  # rows per strip; 8 keeps the 8x8 work tiles of the kernel whole and balances the cheap lower half


def strip_rows(H, world, rank, strip_h=STRIP_H):
    """Frame rows owned by `rank`: strips rank, rank+world, ... of strip_h rows each (last strip may be partial)."""
    ys = np.arange(H)
    return ys[(ys // strip_h) % world == rank]


class _Slot:
    """Buffers (and, on a GPU, the stream) of one frame in flight."""
    __slots__ = ("stream", "local", "slab", "gathered", "frame_buffer")


class FrameRenderer:
    """Renders whole frames of one camera on `world` GPUs (world == 1: plain single-GPU rendering).

    frames_in_flight == 1: render() is asynchronous on torch's current stream; after it returns on rank 0,
    `self.frame_buffer` (H x W x 3 float32, device) holds the sum over `passes` passes once that stream is synchronised.

    frames_in_flight == F > 1: consecutive render() calls rotate over F sets of buffers, each with its own HIP stream, so
    the end of one frame's launch (a few hundred microseconds in which the GPU drains its last paths) and its RCCL gather
    overlap the next frame's rendering.  render() returns the buffer of the frame it enqueued; it is valid after
    `wait()` (current stream waits for every frame in flight) or a device synchronisation, and is overwritten F calls
    later.  The library keeps one scratch set per stream (include/mgpu.h, mgpu_render_strips_device).
    """

    def __init__(self, scene, frame, W, H, maxPathLength, passes, plane=None, seed=1, rank=0, world=1, device=None,
                 strip_h=STRIP_H, render_local=None, tonemap_local=None, frames_in_flight=1, force_collective=False):
        """render_local(rows, out, pass_base): optional replacement for the device renderer -- fills out[:len(rows)]
        (float32, len(rows) x W x 3) for the given frame rows.  Used by the CPU (gloo) tests of the partition/gather
        logic, where `device` is torch.device("cpu"); the product path leaves it None and renders through the C ABI."""
        # force_collective: run the gather and the re-interleaving even with world == 1 (needs an initialised process
        # group): lets ONE GPU exercise the N > 1 code path -- RCCL on side streams, frames in flight -- end to end
        self.collective = world > 1 or force_collective
        self.render_local = render_local
        self.tonemap_local = tonemap_local  # (float strips, passes, mode, out u8) -> None; CPU tests only
        self._ldr = {}                      # per display mode: local / gathered / assembled 8-bit buffers
        self.scene, self.frame = scene, np.ascontiguousarray(frame, "<f8")
        self.W, self.H, self.mpl, self.passes = W, H, maxPathLength, passes
        self.plane = None if plane is None else np.ascontiguousarray(plane, "<f4")
        self.seed, self.rank, self.world, self.strip_h = seed, rank, world, strip_h
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = device
        self.rows = strip_rows(H, world, rank, strip_h)
        self.n_rows = len(self.rows)
        counts = [len(strip_rows(H, world, r, strip_h)) for r in range(world)]
        self.max_rows = max(counts)
        if frames_in_flight < 1:
            raise ValueError("frames_in_flight must be >= 1")
        self.frames_in_flight = frames_in_flight
        if self.collective and rank == 0:
            # frame row y lives at slab row perm[y] = owner(y) * max_rows + local index of y at its owner
            perm = np.empty(H, np.int64)
            for r in range(world):
                rows = strip_rows(H, world, r, strip_h)
                perm[rows] = r * self.max_rows + np.arange(len(rows))
            self.perm = torch.from_numpy(perm).to(self.device)
        self._slots, self._next = [], 0
        for _ in range(frames_in_flight):
            sl = _Slot()
            sl.stream = (torch.cuda.Stream(self.device)
                         if frames_in_flight > 1 and self.device.type == "cuda" else None)
            # local strips, padded to the largest share so the gather is uniform
            sl.local = torch.zeros((self.max_rows, W, 3), dtype=torch.float32, device=self.device)
            sl.slab, sl.gathered = None, None
            if self.collective:
                if rank == 0:
                    # one contiguous landing area [world, max_rows, W, 3]; rank r's padded strips arrive in slab r
                    sl.slab = torch.empty((world, self.max_rows, W, 3), dtype=torch.float32, device=self.device)
                    sl.gathered = list(sl.slab.unbind(0))
                    sl.frame_buffer = torch.empty((H, W, 3), dtype=torch.float32, device=self.device)
                else:
                    sl.frame_buffer = None
            else:
                sl.frame_buffer = sl.local  # n_rows == H
            self._slots.append(sl)
        self._bind(self._slots[0])

    def _bind(self, sl):
        """The attributes of the frame enqueued last (all there is when frames_in_flight == 1)."""
        self.local, self.slab, self.gathered, self.frame_buffer = sl.local, sl.slab, sl.gathered, sl.frame_buffer

    def _take_slot(self):
        sl = self._slots[self._next]
        self._next = (self._next + 1) % self.frames_in_flight
        self._bind(sl)
        return sl

    def wait(self):
        """Makes torch's current stream wait for every frame in flight (no-op when frames_in_flight == 1)."""
        for sl in self._slots:
            if sl.stream is not None:
                torch.cuda.current_stream(self.device).wait_stream(sl.stream)

    def _render_strips(self, pass_base):
        """Float strips of this rank into self.local, on the current stream."""
        if self.render_local is not None:
            if self.n_rows:
                self.render_local(self.rows, self.local, pass_base)
        elif self.n_rows:
            stream = torch.cuda.current_stream(self.device).cuda_stream
            self.scene.render_strips_device(self.frame, self.W, self.H, self.local.data_ptr(), self.n_rows,
                                            y_first=self.rank * self.strip_h, strip_h=self.strip_h,
                                            y_period=self.strip_h * self.world, maxPathLength=self.mpl,
                                            passes=self.passes, plane=self.plane, rng_mode=mgpu.RNG_HASH, seed=self.seed,
                                            pass_base=pass_base, stream=stream)

    def render(self, pass_base=0):
        import contextlib
        import torch.distributed as dist
        sl = self._take_slot()
        if sl.stream is not None:
            sl.stream.wait_stream(torch.cuda.current_stream(self.device))  # whatever the caller enqueued comes first
        with (torch.cuda.stream(sl.stream) if sl.stream is not None else contextlib.nullcontext()):
            self._render_strips(pass_base)
            if self.collective:
                # collectives are issued in frame order on every rank; RCCL runs them on its own stream, ordered after
                # this frame's kernel and before whatever this slot's stream does next
                dist.gather(self.local, self.gathered if self.rank == 0 else None, dst=0)
                if self.rank == 0:  # re-interleave the strips: one gather kernel over rows
                    torch.index_select(self.slab.view(self.world * self.max_rows, self.W, 3), 0, self.perm,
                                       out=self.frame_buffer)
        return self.frame_buffer

    # ---- display frame: only 8-bit pixels leave the GPUs ------------------------------------------
    def _ldr_buffers(self, mode):
        if mode in self._ldr:
            return self._ldr[mode]
        ch = 3 if mode == mgpu.TONEMAP_LINEAR_RGB8 else 4
        b = dict(ch=ch, local=torch.zeros((self.max_rows, self.W, ch), dtype=torch.uint8, device=self.device),
                 count=torch.full((self.max_rows, self.W), self.passes, dtype=torch.int32, device=self.device))
        if self.collective and self.rank == 0:
            b["slab"] = torch.empty((self.world, self.max_rows, self.W, ch), dtype=torch.uint8, device=self.device)
            b["gathered"] = list(b["slab"].unbind(0))
            b["frame"] = torch.empty((self.H, self.W, ch), dtype=torch.uint8, device=self.device)
        elif not self.collective:
            b["frame"] = b["local"]
        self._ldr[mode] = b
        return b

    def render_ldr(self, mode=None, pass_base=0):
        """Renders the frame and returns it DISPLAY-READY on rank 0 (None elsewhere): every rank divides its strips by the
        pass count and applies the driver's transform itself (mode TONEMAP_LINEAR_RGB8: HDRToLDR of main_console.cc:25-43,
        H x W x 3; TONEMAP_GAMMA22_BGRA8: Display of main_sdl.cc:420-477, H x W x 4), and the one gather moves 8-bit
        pixels -- a quarter (RGB8) or a third (BGRA8) of the float traffic.  Pixel for pixel the result equals the
        transform applied to the gathered float frame: it is per-pixel and count = passes everywhere."""
        import torch.distributed as dist
        mode = mgpu.TONEMAP_LINEAR_RGB8 if mode is None else mode
        b = self._ldr_buffers(mode)
        # 1. float strips into self.local, exactly as render() does, but without the float gather (display frames run on
        #    the current stream, one at a time: the 8-bit buffers are not rotated)
        self.wait()
        self._bind(self._slots[0])
        self._render_strips(pass_base)
        # 2. per-rank display transform of the local strips
        if self.n_rows:
            if self.tonemap_local is not None:
                self.tonemap_local(self.local[: self.n_rows], self.passes, mode, b["local"][: self.n_rows])
            else:
                stream = torch.cuda.current_stream(self.device).cuda_stream
                mgpu.tonemap_device(self.local.data_ptr(), b["count"].data_ptr(), self.n_rows * self.W, mode,
                                    b["local"].data_ptr(), device=self.device.index or 0, stream=stream)
        # 3. one gather of 8-bit strips, re-interleaved on rank 0
        if self.collective:
            dist.gather(b["local"], b["gathered"] if self.rank == 0 else None, dst=0)
            if self.rank == 0:
                torch.index_select(b["slab"].view(self.world * self.max_rows, self.W, b["ch"]), 0, self.perm, out=b["frame"])
                return b["frame"]
            return None
        return b["frame"]
