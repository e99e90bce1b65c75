#!/usr/bin/env python3
"""Headline benchmark: ORB extract + projection match, frames/s, KITTI-shaped frames.

One "step" = one pass of the hot path over one batch of synthetic frames resident in HBM:
  extract (pyramid, FAST + NMS, quadtree, orientation, blur, rotated BRIEF) of BATCH + 1 frames of
  1241 x 376 with 2000 keypoints / 8 levels / scale 1.2, then
  projection::match_current_and_last_frames of each of the BATCH frames against its predecessor
  (frame 0 of the batch is the halo frame a shard re-extracts instead of communicating).
`value` = frames matched per second over all ranks (BATCH x steps x ranks / max-over-ranks time).

Multi-GPU: offline frame batches are independent, so ranks process disjoint batches with no
data-path collective ("weak" scaling); torch.distributed (RCCL) is only used for the timing barrier
and the max over ranks.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

from openvslam_amd import _abi as A  # noqa: E402
from openvslam_amd import _lib  # noqa: E402
from openvslam_amd import synth  # noqa: E402
from openvslam_amd.feature import orb_extractor, orb_params  # noqa: E402

WIDTH, HEIGHT, MAX_KP, LEVELS, SCALE = 1241, 376, 2000, 8, 1.2
SHIFT_X, SHIFT_Y = 3, 1
MARGIN = 20.0

# Algorithmic bytes per KITTI frame, stage by stage (each stage reads its input once and writes
# its output once; patch gathers re-touch lines already counted):
#   A = sum of level pixels = 1 444 097, P = level-0 pixels = 466 616, p7 = last level = 36 330
LEVEL_PX = [1241 * 376, 1034 * 313, 862 * 261, 718 * 218, 598 * 181, 499 * 151, 416 * 126, 346 * 105]
A_PX, P_PX, P7_PX = sum(LEVEL_PX), LEVEL_PX[0], LEVEL_PX[-1]
STAGE_BYTES = {
    "resize": (A_PX - P7_PX) + (A_PX - P_PX),            # read levels 0..6, write levels 1..7
    "k_fast_cells": A_PX,                                # read every level once
    "k_gauss7": 2 * A_PX,                                # read + write every level
    "k_quadtree": MAX_KP * 28,                           # keypoints out
    "k_orient_desc": MAX_KP * 32,                        # descriptors out
    "match": 2 * MAX_KP * 32 + MAX_KP * 28,              # two descriptor sets + keypoints in
}
PATH_BYTES_PER_FRAME = sum(STAGE_BYTES.values())         # 7 021 539
STAGE_OF_KERNEL = {"k_resize_rows": "resize", "k_resize_linear": "resize", "k_batch_setup": "match", "k_grid_build": "match",
                   "k_proj_candidates": "match", "k_proj_resolve": "match"}
assert PATH_BYTES_PER_FRAME == 7021539
HBM_PEAK_GBS = 8000.0


def make_frames_on_device(n_frames, device, seed=0xC0FFEE):
    """Synthetic sequence: a window sliding by (SHIFT_X, SHIFT_Y) per frame over one seeded canvas,
    fresh noise per frame (noise drawn on the device from a seeded generator)."""
    canvas = synth.make_sequence(WIDTH + SHIFT_X * n_frames, HEIGHT + SHIFT_Y * n_frames, 1, seed=seed, noise_sigma=0.0)[0]
    canvas = torch.from_numpy(canvas).to(device)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    frames = torch.empty((n_frames, HEIGHT, WIDTH), dtype=torch.uint8, device=device)
    for t in range(n_frames):
        win = canvas[t * SHIFT_Y:t * SHIFT_Y + HEIGHT, t * SHIFT_X:t * SHIFT_X + WIDTH].float()
        coarse = torch.randn((HEIGHT // 2 + 1, WIDTH // 2 + 1), generator=gen, device=device) * 8.0
        noise = coarse.repeat_interleave(2, 0).repeat_interleave(2, 1)[:HEIGHT, :WIDTH]
        fine = torch.randn((HEIGHT, WIDTH), generator=gen, device=device) * 2.0
        frames[t] = (win + noise + fine).round().clamp(0, 255).to(torch.uint8)
    return frames


class Pipeline:
    """Extract + match of one resident batch through the C ABI (device pointers, one stream)."""

    def __init__(self, device_index, n_frames, stream=None):
        self.dev = torch.device("cuda", device_index)
        self.n_frames = n_frames
        self.stream = stream
        self.ext = orb_extractor(orb_params(MAX_KP, SCALE, LEVELS, 20, 7), device=device_index)
        self.cap = self.ext.keypoint_capacity(WIDTH, HEIGHT)
        self.kps = torch.zeros((n_frames, self.cap, 7), dtype=torch.float32, device=self.dev)
        self.desc = torch.zeros((n_frames, self.cap, 32), dtype=torch.uint8, device=self.dev)
        self.counts = torch.zeros((n_frames,), dtype=torch.int32, device=self.dev)
        self.reproj = torch.full((n_frames, self.cap, 2), -1e38, dtype=torch.float32, device=self.dev)
        self.matched = torch.zeros((n_frames, self.cap), dtype=torch.int32, device=self.dev)
        self.num_matches = torch.zeros((n_frames,), dtype=torch.int32, device=self.dev)
        self.shift = torch.tensor([float(SHIFT_X), float(SHIFT_Y)], dtype=torch.float32, device=self.dev)
        # zero-distortion perspective camera: image bounds = image rectangle, 64 x 48 grid
        self.grid = A.make_grid(0.0, 0.0, float(WIDTH), float(HEIGHT))
        self.sf = self.ext.get_scale_factors()
        self.device_index = device_index

    def step(self, frames):
        if self.stream is not None:
            with torch.cuda.stream(self.stream):
                self._step(frames)
        else:
            self._step(frames)

    def _step(self, frames):
        L = _lib.lib()
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        self.ext.extract_batch_dev(frames.data_ptr(), self.n_frames, WIDTH, HEIGHT, WIDTH, WIDTH * HEIGHT,
                                   self.kps.data_ptr(), self.desc.data_ptr(), self.counts.data_ptr(), self.cap, stream)
        # caller-side reprojection (the camera model stays outside the boundary): a point seen at
        # (x, y) in frame f-1 is expected at (x - dx, y - dy) in frame f
        torch.sub(self.kps[:-1, :, 0:2], self.shift, out=self.reproj[1:])
        _lib.check(L.ovs_match_batch_current_and_last_dev(
            self.kps.data_ptr(), self.desc.data_ptr(), self.counts.data_ptr(), self.n_frames, self.cap,
            self.reproj.data_ptr(), C.byref(self.grid), A.ptr(self.sf, A.f32p), LEVELS,
            MARGIN, 1, self.matched.data_ptr(), self.num_matches.data_ptr(), self.device_index, stream))


def _cpu_chunk(frames_host, first, stop, deadline, out, slot):
    """extract + match of frames [first, stop) with frame first - 1 (or first) as the halo."""
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle
    ext = oracle.OrbExtractor(max_num_keypts=MAX_KP, scale_factor=SCALE, num_levels=LEVELS)
    OL = oracle.lib()
    grid = A.make_grid(0.0, 0.0, float(WIDTH), float(HEIGHT))
    sf = oracle.scale_table("scale_factors", LEVELS, SCALE)
    prev = None
    done = 0
    for f in range(first, stop):
        kps, desc = ext.extract(frames_host[f])
        if prev is not None:
            pk, pd = prev
            view = A.make_frame_view(kps, desc, sf, grid)
            reproj = np.stack([pk["x"] - SHIFT_X, pk["y"] - SHIFT_Y], axis=1).astype(np.float32)
            q = A.make_proj_queries(reproj, pk["octave"], pd, angle=pk["angle"])
            matched = np.zeros(len(kps), dtype=np.int32)
            num = C.c_uint32(0)
            oracle.check(OL.ovo_match_current_and_last_frames(view.ref(), q.ref(), None, MARGIN, A.DIR_NONE, LEVELS, 0.0, 1,
                                                              A.ptr(matched, A.i32p), C.byref(num)))
            done += 1
        prev = (kps, desc)
        if time.perf_counter() > deadline and done >= 2:
            break
    out[slot] = done


def cpu_baseline(frames_host, seconds_budget=15.0):
    """The CPU oracle (restated reference path) on a bounded sample of the same frames: first one
    thread (the reference's default build has OpenMP off), then frame-parallel over the host cores
    this process may use (contiguous chunks with a halo frame, as the GPU ranks shard)."""
    import threading
    res = [0]
    t0 = time.perf_counter()
    _cpu_chunk(frames_host, 0, len(frames_host), t0 + seconds_budget / 3.0, res, 0)
    single = res[0] / (time.perf_counter() - t0)
    n_thr = max(1, min(16, len(os.sched_getaffinity(0)), (len(frames_host) - 1) // 4))
    per = (len(frames_host) - 1) // n_thr
    res = [0] * n_thr
    t0 = time.perf_counter()
    deadline = t0 + 2.0 * seconds_budget / 3.0
    threads = [threading.Thread(target=_cpu_chunk, args=(frames_host, i * per, i * per + per + 1, deadline, res, i)) for i in range(n_thr)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    dt = time.perf_counter() - t0
    return sum(res) / dt, sum(res), n_thr, single


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 counter passes (FETCH_SIZE and
    WRITE_SIZE, separate runs, gfx950 corrections applied by tools/summarize_prof.py)."""
    import csv
    path = ROOT / "profiles" / "r1_pmc_per_kernel.csv"
    if not path.exists():
        return None
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["kernel"] == kernel:
                try:
                    return float(row["hbm_read_bytes(=2*FETCH_SIZE*1024)"]) + float(row["hbm_write_bytes(=WRITE_SIZE*1024)"])
                except ValueError:
                    return None
    return None


def stream_copy_gbs(dev):
    """Measured device-to-device copy rate (bytes read + written per second) of a 1 GiB buffer."""
    a = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    b = torch.empty_like(a)
    b.copy_(a)
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize(dev)
    return 5 * 2 * (1 << 30) / (e0.elapsed_time(e1) * 1e-3) / 1e9


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1020, help="frames matched per step and GPU (a multiple of --streams)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=4, help="contiguous sub-batches of the step's batch in flight on separate HIP streams (each re-extracts one halo frame)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    n_frames = args.batch + 1
    frames = make_frames_on_device(n_frames, dev, seed=0xC0FFEE + rank)
    # The batch as `streams` contiguous sub-batches, each with its own extractor handle, buffers and
    # stream; sub-batch k starts with the last frame of sub-batch k - 1 as its halo.  The defaults make
    # every launch work on 255 + 1 = 256 frames: the kernels with one workgroup per (frame, level) or
    # per frame then fill the 256 compute units without a ragged last round, and three streams keep
    # the device busy while such a kernel of one stream runs.
    per = args.batch // args.streams
    if per * args.streams != args.batch:
        raise SystemExit("--batch must be a multiple of --streams")
    pipe = Pipeline(local_rank, per + 1)  # one sub-batch on the default stream: the per-kernel timing pass
    if args.streams > 1:
        subs = [Pipeline(local_rank, per + 1, torch.cuda.Stream(dev)) for _ in range(args.streams)]

        class Multi:
            counts = property(lambda self_: torch.cat([sp.counts[1:] for sp in subs]))
            num_matches = property(lambda self_: torch.cat([subs[0].num_matches[:1] * 0] + [sp.num_matches[1:] for sp in subs]))

            def step(self_, fr):
                for k, sp in enumerate(subs):
                    sp.step(fr[k * per:k * per + per + 1])

        multi = Multi()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    timed = multi if args.streams > 1 else pipe
    for _ in range(args.warmup):
        timed.step(frames)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        timed.step(frames)
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    barrier()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    counts = timed.counts.cpu().numpy()
    nm = timed.num_matches.cpu().numpy()
    value = args.batch * args.steps * world / elapsed

    out = {
        "metric": "ORB extract+match frames/sec @KITTI 1241x376 2k kp",
        "value": value,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": "KITTI-shaped 1241x376 mono batch, 2000 kp, 8 levels, scale 1.2, "
                               "extract + match::projection (current vs last frame, margin 20, orientation check)",
                   "frames_per_step_per_gpu": args.batch, "halo_frames_per_step": args.streams, "streams": args.streams,
                   "frames_per_launch": per + 1,
                   "mean_keypoints_per_frame": float(counts.mean()), "mean_matches_per_frame": float(nm[1:].mean())},
    }

    if rank == 0:
        # per-kernel device time of the same step (separate, untimed pass with event pairs around
        # every launch on the launch stream; the whole batch on ONE stream, so that no other kernel
        # shares the device while a launch is being timed)
        L = _lib.lib()
        L.ovs_timing_reset()
        L.ovs_timing_enable(1)
        prof_steps = 3
        for _ in range(prof_steps):
            pipe.step(frames[:per + 1])
        torch.cuda.synchronize(dev)
        L.ovs_timing_enable(0)
        rep = _lib.timing_report()
        per_step = {k: v[0] / prof_steps for k, v in rep.items()}
        launches = {k: v[1] // prof_steps for k, v in rep.items()}
        dominant = max(per_step, key=per_step.get)
        bytes_key = STAGE_OF_KERNEL.get(dominant, dominant)
        alg_bytes_per_launch = STAGE_BYTES[bytes_key] * (per + 1) / max(launches[dominant], 1)
        avg_ms = per_step[dominant] / max(launches[dominant], 1)
        achieved = alg_bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        out["roofline"] = {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(dominant),
                           "traffic_source": "profiles/r1_pmc_per_kernel.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)",
                           "stream_copy_gbs": stream_copy_gbs(dev),
                           "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": alg_bytes_per_launch,
                           "path_bytes_per_frame": PATH_BYTES_PER_FRAME,
                           "path_achieved_gbs": PATH_BYTES_PER_FRAME * value / world / 1e9,
                           "path_frac": PATH_BYTES_PER_FRAME * value / world / 1e9 / HBM_PEAK_GBS,
                           "kernel_ms_per_launch_set": per_step,
                           "timing_pass": "one sub-batch of %d frames alone on one stream" % (per + 1)}
        if world == 1 and not args.no_cpu_baseline:
            host_frames = frames[:min(n_frames, 257)].cpu().numpy()
            fps, sample, cores, single = cpu_baseline(host_frames, args.cpu_seconds)
            out["cpu_baseline"] = {"value": fps, "unit": "frames/s", "cores": cores, "kind": "port",
                                   "single_thread_value": single,
                                   "sample": "%d frames of the same synthetic sequence in %d contiguous chunks (one halo frame each), "
                                             "extract + match, one oracle thread per chunk" % (sample, cores)}
        print(json.dumps(out))

    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
