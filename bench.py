#!/usr/bin/env python3
"""bench.py — the path's headline metric on MI355X (BASELINE.json: encode+decode Mpixels/s, Kodak, K=4 N=192).

    python bench.py --gpus N --steps K --warmup W

Workload (config.workload = "kodak24"): BASELINE.json configs[1] — 24 Kodak-sized images (768x512 ->
y [1,192,32,48] -> two checkerboard halves [1,192,32,24] each), mixture parameters [1,768,32,24] x3 per half,
synthetic and seeded (no Kodak files / checkpoint exist offline), resident in HBM before the
timed region.  One STEP = one pass of the hot path over that batch:
    GaussianMixtureConditional.compress  on all 48 halves   (quant_stats + symtab kernels, D2H tables, host rANS)
    GaussianMixtureConditional.decompress on all 48 halves  (cdftab kernel, D2H tables, host rANS, H2D + scatter)
value = pixels of all images of all ranks / wall time (Mpixels/s).  N > 1: one process per GPU, each rank codes its
own 24 images (weak scaling); the only exchange is one RCCL all-gather of the per-stream byte lengths per step.

One JSON line is printed by rank 0; besides the driver's contract it carries
  roofline      the symtab (encode-side GMM-CDF) kernel: algorithmic bytes (56 B/coded symbol) over
                its launch duration measured here with HIP events on the stream it runs on, against 8 TB/s HBM
  cpu_baseline  the REAL reference extension (oracle/_ref, built from /root/reference in the build container;
                kind "reference"), or this repo's C restatement (kind "port"), timed on one host core of this box
                on the same 24 images
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
# algorithmic bytes per coded symbol of the symtab kernel: 4 (y) + 3*4*{4|2} (sigma, mu, pi) in + 4 out


ELIC_GROUPS = (16, 16, 32, 64, 192)  # elic_gmm.py:92-96


def make_workload(rank: int, images: int, dev, workload: str = "kodak24", f16: bool = False):
    """-> (host arrays per stream, device tensors per stream, pixels per image).
    kodak24: 2 streams per image, [1,192,32,24]; elic4k: 10 streams per image (5 channel groups x 2 halves of a
    3840x2160 image padded to 2176 rows -> y [1,320,136,240], SURVEY.md §8 sizes)."""
    from flashgmm_amd import testing as T

    host, devt = [], []
    if workload == "kodak24":
        shapes = [(192, 32, 24)] * 2
        pix = 768 * 512
    else:
        shapes = [(g, 136, 120) for g in ELIC_GROUPS for _ in range(2)]
        pix = 3840 * 2160
    for i in range(images):
        for j, (M, h, w) in enumerate(shapes):
            y, sg, mu, pi = T.make_latent(1000 * rank + i * len(shapes) + j, M=M, h=h, w=w)  # sigma pre-clamped as KA-1
            if f16:
                sg, mu, pi = T.to_float16_planes(sg, mu, pi)
            host.append((y, sg, mu, pi))
            devt.append([torch.from_numpy(a).to(dev) for a in (y, sg, mu, pi)])
    return host, devt, pix


def cpu_baseline(host, pix_per_image: int, streams_per_image: int, budget_s: float = 12.0):
    """Time the reference's own coder on ONE core of this box on the same images (bounded sample)."""
    from flashgmm_amd import testing as T
    from oracle import oracle as O

    prepared = []
    for y, sg, mu, pi in host:
        sym, s, m, w, abs_max, zb, yq = T.to_coder_inputs(y, *(a.astype(np.float32) for a in (sg, mu, pi)))
        prepared.append((sym, s, m, w, abs_max))
    kind = "port"
    enc = dec = None
    if O.ref_available():
        try:
            os.environ.pop("APPROX_MODE", None)
            ans = O.ref_ans()
            kind = "reference"
            # the reference is handed (n,4) views with strides (1,n) (entropy_models.py:810-828)
            tens = [(torch.from_numpy(sym), *(torch.from_numpy(np.ascontiguousarray(a.T)).T for a in (s, m, w)), am)
                    for sym, s, m, w, am in prepared]

            def enc(i):
                sym, s, m, w, am = tens[i]
                return ans.RansEncoder().encode_with_indexes_gmm(sym, s, m, w, am + 1)

            def dec(i, b):
                sym, s, m, w, am = tens[i]
                return ans.RansDecoder().decode_with_indexes_gmm(b, s, m, w, am + 1).numpy()
        except Exception as e:  # pragma: no cover - e.g. ISA mismatch on an unexpected host
            print(f"[bench] reference extension unusable here ({e}); falling back to the C restatement", file=sys.stderr)
            kind = "port"
    if kind == "port":
        def enc(i):
            sym, s, m, w, am = prepared[i]
            return O.encode_gmm(0, sym, s, m, w)

        def dec(i, b):
            sym, s, m, w, am = prepared[i]
            return O.decode_gmm(0, b, s, m, w, am + 1)

    torch.set_num_threads(1)
    best = None
    passes = 0
    t_start = time.perf_counter()
    while passes < 5 and (passes < 1 or time.perf_counter() - t_start < budget_s):
        t0 = time.perf_counter()
        for i in range(len(prepared)):
            b = enc(i)
            d = dec(i, b)
        dt = time.perf_counter() - t0
        assert np.array_equal(d, prepared[-1][0])
        best = dt if best is None else min(best, dt)
        passes += 1
    n_img = len(prepared) // streams_per_image
    n_sym = sum(len(p[0]) for p in prepared)
    return {
        "value": round(n_img * pix_per_image / best / 1e6, 3),
        "unit": "Mpixels/s",
        "cores": 1,
        "kind": kind,
        "sample": f"{n_img} image(s) x {streams_per_image} streams ({n_sym} symbols), encode+decode, best of {passes} passes, "
                  f"{best * 1e3:.0f} ms/pass = {best / n_sym * 1e9:.0f} ns/symbol",
    }


def pmc_traffic(workload: str, mode: str, f16: bool):
    """roofline.traffic: HBM bytes per symtab launch from rocprofv3 PMC passes (scripts/collect_pmc.sh, committed under
    profiles/), gfx950-corrected as MI355X_MICROARCH.md prescribes.  PMC collection needs the profiler, so bench.py
    reports the committed measurement of this same workload, or null when there is none."""
    if workload != "kodak24" or f16:
        return None
    best = None
    for f in sorted(__import__("glob").glob(os.path.join(ROOT, "profiles", "r*_pmc_symtab.json"))):
        try:
            d = json.load(open(f))
            if d.get("workload") == workload and d.get("mode") == mode:
                best = d["symtab"]["hbm_bytes_corrected"]
        except Exception:
            pass
    return best


def launch_ranks(a, argv):
    """`python bench.py --gpus N` with N > 1 and no torch.distributed.run around it: start N fresh child processes, one
    per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment), BEFORE anything in this process touches the
    GPU; forward rank 0's JSON line; exit non-zero if any rank fails.  Children are started as children (never exec'ed
    over this process)."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), FGMM_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0, _ = procs[0].communicate()
    rcs = [p.wait() for p in procs]
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    if any(rcs):
        raise SystemExit(f"bench.py: rank exit codes {rcs}")


_REAL_STDOUT = None


def emit(line: dict) -> None:
    out = _REAL_STDOUT or sys.stdout
    out.write(json.dumps(line) + "\n")
    out.flush()


def dryrun(a, world, rank):
    """FGMM_BENCH_DRYRUN=1 (tests/test_bench_launcher_cpu.py): everything of an N-rank run EXCEPT the GPU work — process
    group (gloo), the per-step all-gather of stream lengths, barriers, max-over-ranks timing, rank 0's one JSON line.
    The line says so ("data": "dryrun-no-gpu", value 0): it is never a measurement."""
    import torch.distributed as dist
    from flashgmm_amd import parallel as P

    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    n_streams = 2 * (a.images or 24)
    t_gather = []

    def step():
        lens = [1000 + 7 * rank + i for i in range(n_streams)]  # stand-in for the coder's output lengths
        t0 = time.perf_counter()
        g = P.all_gather_stream_lengths(lens, n_streams) if world > 1 else None
        t_gather.append(time.perf_counter() - t0)
        return g

    for _ in range(max(a.warmup, 1)):
        g = step()
    if world > 1:
        assert g.shape == (world, n_streams) and g[rank].tolist() == [1000 + 7 * rank + i for i in range(n_streams)]
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    per_rank = [dt]
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64)
        gathered = [torch.empty_like(tt) for _ in range(world)]
        dist.all_gather(gathered, tt)
        per_rank = [float(t) for t in gathered]
        dt = max(per_rank)
    if rank == 0:
        emit(({"metric": "dryrun", "value": 0.0, "unit": "Mpixels/s", "n_gpus": world, "steps": a.steps,
                          "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "dryrun-no-gpu",
                          "config": {"workload": a.workload, "streams_per_gpu": n_streams},
                          "ranks": {"backend": "gloo" if world > 1 else None, "ms_per_step": [round(t / a.steps * 1e3, 3) for t in per_rank],
                                    "allgather_ms": round(float(np.mean(t_gather[-a.steps:])) * 1e3, 4)}}))
    if world > 1:
        dist.destroy_process_group()


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="kodak24", choices=["kodak24", "elic4k"])
    ap.add_argument("--images", type=int, default=None, help="images per GPU (default 24 for kodak24, 1 for elic4k)")
    ap.add_argument("--param-dtype", default=None, choices=["f32", "f16"], help="default f32 (kodak24) / f16 (elic4k)")
    ap.add_argument("--mode", default="polya", choices=["polya", "as", "logistic"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args(argv)

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(a, argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {a.gpus}, or without a launcher")
    # stdout carries exactly ONE line (rank 0's JSON): whatever libraries print there (gloo / RCCL banners) goes to stderr
    global _REAL_STDOUT
    sys.stdout.flush()
    _REAL_STDOUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    if os.environ.get("FGMM_BENCH_DRYRUN"):  # no GPU: the launcher, the collective and the JSON contract only (CPU test)
        return dryrun(a, world, rank)
    if os.environ.get("FGMM_BENCH_ONE_DEVICE"):  # rehearsal of the N > 1 code path on a 1-GPU box (dev aid)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("FGMM_BENCH_ONE_DEVICE"):  # RCCL refuses two ranks on one GPU: rehearse with gloo
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    coll_dev = dev if (dist and dist.get_backend() == "nccl") else torch.device("cpu")

    from flashgmm_amd import GaussianMixtureConditional, _lib
    from flashgmm_amd import parallel as P

    numa = P.bind_to_gpu_numa_node(local_rank)  # before the worker threads and pinned buffers exist

    if a.images is None:
        a.images = 24 if a.workload == "kodak24" else 1
    f16 = (a.param_dtype or ("f32" if a.workload == "kodak24" else "f16")) == "f16"
    host, devt, pix_per_image = make_workload(rank, a.images, dev, a.workload, f16)
    streams_per_image = len(devt) // a.images
    bytes_per_symbol = 32 if f16 else 56
    ys = [t[0] for t in devt]
    ss = [t[1] for t in devt]
    ms = [t[2] for t in devt]
    ws = [t[3] for t in devt]
    shapes = sorted({tuple(y.shape) for y in ys})
    n_streams = len(ys)
    hw_of = [y.shape[2] * y.shape[3] for y in ys]
    stacked = len(shapes) == 1
    if stacked:  # items of one shape go in as ONE tensor each, [N, ., h, w]: what a network run on a batch produces
        ys, ss, ms, ws = (torch.cat(t) for t in (ys, ss, ms, ws))
    gmc = GaussianMixtureConditional(K=4, mode=a.mode)
    _lib.set_profiling(local_rank, True)

    k_sym, k_tab, k_fill, k_qs = [], [], [], []

    def step(record=False):
        res = gmc.compress_batch(ys, ss, ms, ws)
        if record:
            k_sym.append(_lib.kernel_ms(local_rank, 0))
            k_qs.append(_lib.kernel_ms(local_rank, 2))
        if world > 1:  # the path's one exchange: per-stream bitstream lengths, RCCL all-gather
            P.all_gather_stream_lengths([len(r[0][0]) for r in res], len(res), device=coll_dev)
        outs = gmc.decompress_batch([r[0][0] for r in res], [r[0][1] for r in res], [r[0][2] for r in res], ss, ms, ws)
        if record:
            k_tab.append(_lib.kernel_ms(local_rank, 1))
        return res, outs

    for _ in range(max(a.warmup, 1)):
        res, outs = step()
    # correctness of what is being timed: decode(encode(y)) == round(y) for every stream of this rank
    for i in range(n_streams):
        y_i = ys[i:i + 1] if stacked else ys[i]
        assert torch.equal(outs[i], res[i][1]) and torch.equal(res[i][1], torch.round(y_i)), f"stream {i} mismatch"
    n_coded = sum(int(r[0][2].sum()) * hw for r, hw in zip(res, hw_of))
    total_bytes = sum(len(r[0][0]) for r in res)

    # a generational GC pass of the interpreter (tens of ms with torch loaded) is not part of the path
    import gc

    gc.collect()
    gc.freeze()
    gc.disable()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step(record=True)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    if dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        ms_per_step = dt / a.steps * 1e3
        value = world * a.images * pix_per_image * a.steps / dt / 1e6
        sym_ms = float(np.mean(k_sym))
        achieved = n_coded * bytes_per_symbol / (sym_ms * 1e-3) / 1e9
        out = {
            # BASELINE.json's metric, verbatim: `value` is its Mpixels/s half, the `roofline` object its GMM-CDF HBM half
            "metric": "encode+decode Mpixels/s (Kodak, K=4 N=192) + GMM-CDF HBM GB/s vs roofline" if a.workload == "kodak24"
            else "encode+decode Mpixels/s (ELIC 4K, K=4, fp16 params)",
            "value": round(value, 2),
            "unit": "Mpixels/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",  # the CDF arithmetic; parameter planes: config.param_dtype
            "data": "synthetic",
            "config": {"workload": a.workload, "images_per_gpu": a.images, "streams_per_gpu": n_streams,
                       "stream_shapes": shapes, "stacked_input": stacked, "K": 4, "approx_mode": a.mode,
                       "param_dtype": "f16" if f16 else "f32",
                       "coded_symbols_per_gpu": n_coded, "bitstream_bytes_per_gpu": total_bytes,
                       "host_threads_per_gpu": _lib.lib().fgmm_ctx_threads(_lib.ctx(local_rank)), "numa": numa,
                       "parallelism": f"images sharded over {world} GPU(s)"},
            "roofline": {"bound": "hbm", "kernel": "symtab_kernel (encode-side GMM-CDF)",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(a.workload, a.mode, f16),
                         "launch_ms": round(sym_ms, 4), "bytes_per_launch": n_coded * bytes_per_symbol,
                         "bytes_per_symbol": bytes_per_symbol},
            "kernels_ms": {"symtab": round(sym_ms, 4), "tab_kernels_all_launches": round(float(np.mean(k_tab)), 4), "quant_stats": round(float(np.mean(k_qs)), 4)},
            # what crosses PCIe per step and rank (the decode-side tables are the longest leg of a step)
            "pcie": {"encode_tables_bytes": _lib.ctx_stat(local_rank, 0), "decode_tables_bytes": _lib.ctx_stat(local_rank, 1),
                     "decode_table_bytes_per_latent": round(_lib.ctx_stat(local_rank, 1) / max(1, _lib.ctx_stat(local_rank, 2)), 2),
                     "bitstream_bytes": total_bytes},
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(host, pix_per_image, streams_per_image)
        emit(out)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
