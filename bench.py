#!/usr/bin/env python3
"""Headline benchmark: ESRGAN RRDBNet x4 bf16 inference throughput on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

A step is one GeneratorRRDB forward over one batch of synthetic LR patches already resident in
HBM (BASELINE.json configs[1]: 16 x 3x128x128 -> 16 x 3x512x512, 23 RRDB, bf16 kernels).  With
N > 1 (launched by torch.distributed.run, one rank per GPU) every rank runs its own batch — the
path shards by independent patches with no data-path collective — and the value is the job
aggregate over max-over-ranks time ("weak" scaling).  Rank 0 prints ONE JSON line.

Besides the throughput the line carries
* ``roofline``: for the conv3x3 MFMA kernel instantiation with the largest time share, the
  algorithmic FLOPs of its launches in one forward divided by their measured durations
  (HIP events on the launch stream), against the 2.5 PFLOP/s dense bf16 peak;
* ``cpu_baseline``: the CPU oracle (a port of the reference PyTorch forward) timed on this
  box's host cores on BASELINE configs[0] (batch 4 of 3x64x64);
* ``other_configs``: the remaining BASELINE configurations (510x339 image whole / tiled / sharded,
  CycleGAN GeneratorSR, conditional VAE), ``strong_scaling`` (N > 1: the headline batch split over the
  ranks) and ``train`` (pixel and adversarial iterations) as extras.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0       # MI355X dense bf16 MFMA peak (no sparsity)
PEAK_HBM_GBPS = 8000.0          # MI355X HBM3E peak
FLOP_PER_LR_PIXEL = 35_853_696  # 2 * 17,926,848 MAC for the default generator


def conv_flops(cin, cout, pixels):
    return 2.0 * 9 * cin * cout * pixels


def measure_kernels(net, N, H, W, forwards=3):
    """Time every distinct conv launch configuration of one forward on cfg-sized buffers and
    aggregate per kernel instantiation (Cout<=32 -> MB=1 symbol, Cout 64 -> MB=2 symbol).  Each shape is
    launched ``forwards`` x its per-forward count, so the launch mix of this pass equals that of the timed
    forwards and a profiler's per-symbol average over the whole run is the same average."""
    import super_resolution_amd as S
    from super_resolution_amd import ops
    dev = torch.device("cuda", torch.cuda.current_device())
    dt = S.DTYPE_BF16
    nf, gc, nb = net.num_features, net.growth_channels, net.num_blocks
    rdb = net.body[0].rdb1
    packed = [ops.pack_weight(c.weight.detach(), c.bias.detach(), dt) for c in rdb._convs()]
    buf = torch.zeros((N, H, (nf + 4 * gc) // 8, W, 8), dtype=torch.bfloat16, device=dev).normal_()
    nxt = torch.empty_like(buf)
    trunk = torch.zeros((N, H, nf // 4, W, 4), dtype=torch.float32, device=dev).normal_()
    outer = trunk.clone()
    P = N * H * W

    def timed(count, *conv, **kw):
        """Average duration of one launch: the argument record is built once, then the C entry point is
        called back to back on torch's current stream between two HIP events recorded on that stream."""
        a = ops.conv_args(*conv, **kw)
        reps = count * forwards
        ops.launch_conv(a)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.launch_conv(a)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / reps

    rows = []   # (symbol, name, launches per forward, flops per launch, seconds per launch)
    for k in range(4):
        wp, bp = packed[k]
        cu = (nf + k * gc) // 8
        tk = timed(3 * nb, buf, cu, wp, bp, gc, 0.2, 0, None, 0.0, None, 0.0, buf, cu, None, None)
        rows.append(("MB1", f"conv{k + 1} {nf + k * gc}->{gc}", 3 * nb, conv_flops(nf + k * gc, gc, P), tk,
                     conv_bytes(nf + k * gc, gc, P)))
    wp, bp = packed[4]
    cu = (nf + 4 * gc) // 8
    if net.dense_skip_fp32:          # an fp32 carrier for every skip
        t5 = timed(2 * nb, buf, cu, wp, bp, nf, 1.0, 0, trunk, 0.2, None, 0.0, nxt, 0, trunk, None)
        b5 = conv_bytes(nf + 4 * gc, nf, P, skips=1, f32_out=True)
        t5b = timed(nb, buf, cu, wp, bp, nf, 1.0, 0, trunk, 0.2, outer, 0.2, nxt, 0, outer, None)
        b5b = conv_bytes(nf + 4 * gc, nf, P, skips=2, f32_out=True)
    else:                            # dense-block skips read the bf16 activations; fp32 carrier for the RRDB skip only
        t5 = timed(2 * nb, buf, cu, wp, bp, nf, 1.0, 0, None, 0.2, None, 0.0, nxt, 0, None, None, res_act=buf)
        b5 = conv_bytes(nf + 4 * gc, nf, P, bf16_skips=1)
        t5b = timed(nb, buf, cu, wp, bp, nf, 1.0, 0, None, 0.2, outer, 0.2, nxt, 0, outer, None, res_act=buf)
        b5b = conv_bytes(nf + 4 * gc, nf, P, skips=1, f32_out=True, bf16_skips=1)
    rows.append(("MB2", f"conv5 {nf + 4 * gc}->{nf} +skip", 2 * nb, conv_flops(nf + 4 * gc, nf, P), t5, b5))
    rows.append(("MB2", f"conv5 {nf + 4 * gc}->{nf} +2 skips", nb, conv_flops(nf + 4 * gc, nf, P), t5b, b5b))
    # HR side (64 -> 64 at 2x / 4x resolution, 64 -> 3 at 4x)
    # HR-side convs as the timed forward runs them: fp16 operands, or bf16 with the weight rounding residue (2x MFMA work)
    f16 = bool(net.hr_f16)
    sp = 0 if f16 else (1 if net.hr_split else 0)
    mode = 3 if f16 else sp          # SR355_PACK_F16 / SR355_SPLIT_*
    hr = dict(weight_split=sp, f16_in=f16, f16_out=f16)
    hs1, hs2 = ("MB1f16", "MB2f16") if f16 else ("MB1", "MB2")     # the fp16 instantiations are kernels of their own
    wt, bt = ops.pack_weight(net.trunk_conv.weight.detach(), net.trunk_conv.bias.detach(), dt)
    w64, b64 = ops.pack_weight(net.HR_conv.weight.detach(), net.HR_conv.bias.detach(), dt, mode)
    wl, bl = ops.pack_weight(net.conv_last.weight.detach(), net.conv_last.bias.detach(), dt, mode)
    lr = torch.zeros((N, H, nf // 8, W, 8), dtype=torch.bfloat16, device=dev).normal_()
    h1 = torch.empty((N, 2 * H, nf // 8, 2 * W, 8), dtype=torch.bfloat16, device=dev)
    h2 = torch.empty((N, 4 * H, nf // 8, 4 * W, 8), dtype=torch.bfloat16, device=dev)
    h3 = torch.empty_like(h2)
    out = torch.empty((N, net.out_channels, 4 * H, 4 * W), dtype=torch.float32, device=dev)
    nu = nf // 8
    ta = timed(1, lr, nu, w64, b64, nf, 0.2, 1, None, 0.0, None, 0.0, h1, 0, None, None, **hr)
    rows.append((hs2, "upconv1 (x2 fused)", 1, conv_flops(nf, nf, 4 * P), ta, conv_bytes(nf, nf, 4 * P, P)))
    tb = timed(1, h1, nu, w64, b64, nf, 0.2, 1, None, 0.0, None, 0.0, h2, 0, None, None, **hr)
    rows.append((hs2, "upconv2 (x2 fused)", 1, conv_flops(nf, nf, 16 * P), tb, conv_bytes(nf, nf, 16 * P, 4 * P)))
    tc = timed(1, h2, nu, w64, b64, nf, 0.2, 0, None, 0.0, None, 0.0, h3, 0, None, None, **hr)
    rows.append((hs2, "HR_conv", 1, conv_flops(nf, nf, 16 * P), tc, conv_bytes(nf, nf, 16 * P)))
    td = timed(1, h3, nu, wl, bl, net.out_channels, 1.0, 0, None, 0.0, None, 0.0, None, 0, None, out, weight_split=sp, f16_in=f16)
    rows.append((hs1, "conv_last -> NCHW", 1, conv_flops(nf, net.out_channels, 16 * P), td,
                 conv_bytes(nf, net.out_channels, 16 * P, bf16_out=False, nchw_out=net.out_channels)))
    tt = timed(1, buf, nu, wt, bt, nf, 1.0, 0, trunk, 1.0, None, 0.0, lr, 0, None, None)
    rows.append(("MB2", "trunk_conv +skip", 1, conv_flops(nf, nf, P), tt, conv_bytes(nf, nf, P, skips=1)))
    return rows


def conv_bytes(cin, cout, p_out, p_in=None, skips=0, f32_out=False, bf16_out=True, nchw_out=0, bf16_skips=0):
    """Algorithmic HBM bytes of one launch: bf16 input read once, bf16 output, fp32 skip reads / carrier write,
    skips read from bf16 activations, fp32 NCHW image (weights, a few hundred KB, not counted)."""
    p_in = p_out if p_in is None else p_in
    return (2 * cin * p_in + (2 * cout * p_out if bf16_out else 0) + 4 * cout * p_out * skips +
            2 * cout * p_out * bf16_skips +
            (4 * cout * p_out if f32_out else 0) + 4 * nchw_out * p_out)


def pmc_traffic_file():
    """The newest committed PMC traffic summary under profiles/ (named per round: r03b_... sorts after r02l_...)."""
    try:
        names = sorted(n for n in os.listdir(os.path.join(ROOT, "profiles")) if n.endswith("_pmc_traffic.json"))
    except OSError:
        return None
    return os.path.join(ROOT, "profiles", names[-1]) if names else None


def _traffic_source():
    f = pmc_traffic_file()
    return (f"profiles/{os.path.basename(f)}" if f else "no committed summary") + \
        " (bytes per launch: 2 x FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes over this command; the " \
        "committed summary, not a live reading)"


def measured_traffic(symbol_fragment):
    """HBM-side bytes per launch of a kernel from the committed PMC summary (the bench cannot run the
    profiler on itself); None when the file is absent or has no such kernel."""
    path = pmc_traffic_file()
    try:
        with open(path) as f:
            kernels = json.load(f)["kernels"]
    except (OSError, ValueError, KeyError, TypeError):
        return None
    for name, v in kernels.items():
        if symbol_fragment in name:
            return v["traffic_bytes_per_launch"]
    return None


def chain_roofline(net, x, rows, forwards=10):
    """The persistent launch that runs the generator's body (every dense block and trunk_conv; conv3x3_chain.hip):
    its duration from HIP events the library records on the launch stream right around it (sr355_chain_timing), over
    ``forwards`` forwards of the timed workload, against the algorithmic FLOPs / bytes of the convs it replaces (the
    per-launch figures of ``rows``).  None when the workload does not take that path."""
    from super_resolution_amd import _lib
    L = _lib.lib()
    L.sr355_chain_timing(1)
    try:
        with torch.no_grad():
            net(x)
            torch.cuda.synchronize()
            if L.sr355_chain_last_ms() < 0:
                return None
            ms = []
            for _ in range(forwards):
                net(x)
                ms.append(float(L.sr355_chain_last_ms()))
    finally:
        L.sr355_chain_timing(0)
    sec = sum(ms) / len(ms) * 1e-3
    body = [r for r in rows if r[0] == "MB1" and "conv_last" not in r[1] or r[1].startswith(("conv5", "trunk_conv"))]
    flop = sum(cnt * fl for _s, _n, cnt, fl, _t, _b in body)
    nbytes = sum(cnt * b for _s, _n, cnt, _fl, _t, b in body)
    per_conv = sum(cnt * t for _s, _n, cnt, _fl, t, _b in body)
    ridge = PEAK_BF16_TFLOPS * 1e12 / (PEAK_HBM_GBPS * 1e9)
    mfma = {"achieved": round(flop / sec / 1e12, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
            "frac": round(flop / sec / 1e12 / PEAK_BF16_TFLOPS, 4)}
    hbm = {"achieved": round(nbytes / sec / 1e9, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
           "frac": round(nbytes / sec / 1e9 / PEAK_HBM_GBPS, 4)}
    # The matrix roof is the one the measurement plan names for the conv kernels (target: half the dense bf16 peak), so it
    # is the headline fraction; the memory side of the same launch stands beside it (the launch sits below the ridge of
    # the two peaks, so by byte count the memory roof is the nearer one: both are reported, neither is hidden).
    return {"bound": "mfma",
            "kernel": "conv3x3_chain (the 345 body convs + trunk_conv as one persistent launch)",
            **mfma,
            "traffic": measured_traffic("conv3x3_chain"),
            "traffic_source": _traffic_source(),
            "avg_launch_us": round(sec * 1e6, 1), "launches_per_forward": 1, "flop_per_launch": flop,
            "algorithmic_bytes_per_launch": round(nbytes),
            "flop_per_byte": round(flop / nbytes, 1), "ridge_flop_per_byte": round(ridge, 1),
            "hbm_side": hbm,
            "same_convs_one_launch_each_us": round(per_conv * 1e6, 1),
            "how": f"HIP events recorded by the library on the launch stream right before and after the kernel, mean of "
                   f"{forwards} forwards of the timed workload; FLOPs and bytes are the sums over the convs it runs"}


def roofline_from(rows):
    agg = {}
    for sym, _name, cnt, fl, sec, nbytes in rows:
        a = agg.setdefault(sym, [0, 0.0, 0.0, 0.0])
        a[0] += cnt
        a[1] += cnt * fl
        a[2] += cnt * sec
        a[3] += cnt * nbytes
    out = {}
    for sym, (cnt, fl, sec, nbytes) in agg.items():
        out[sym] = {"launches_per_forward": cnt, "avg_launch_us": sec / cnt * 1e6, "flop_per_launch": fl / cnt,
                    "achieved": fl / sec / 1e12, "time_s": sec, "bytes_per_launch": nbytes / cnt}
    dom = max(out, key=lambda s: out[s]["time_s"])
    names = {"MB1": "conv3x3_fine<MB=1,NB=4,R=2,bf16> (Cout<=32)", "MB2": "conv3x3_fine<MB=2,NB=2,R=2,bf16> (Cout=64)",
             "MB1f16": "conv3x3_fine<MB=1,NB=4,R=2,fp16> (conv_last)", "MB2f16": "conv3x3_fine<MB=2,NB=2,R=2,fp16> (HR side)"}
    d = out[dom]
    symbols = {"MB1": "conv3x3_fine<1, 4, 2, 2, false>", "MB2": "conv3x3_fine<2, 2, 2, 2, false>",
               "MB1f16": "conv3x3_fine<1, 4, 2, 2, true>", "MB2f16": "conv3x3_fine<2, 2, 2, 2, true>"}
    roof = {"bound": "mfma", "kernel": names[dom], "achieved": round(d["achieved"], 2), "peak": PEAK_BF16_TFLOPS,
            "unit": "TFLOP/s", "frac": round(d["achieved"] / PEAK_BF16_TFLOPS, 4),
            "traffic": measured_traffic(symbols[dom]),
            "traffic_source": _traffic_source(),
            "avg_launch_us": round(d["avg_launch_us"], 2), "launches_per_forward": d["launches_per_forward"],
            "flop_per_launch": d["flop_per_launch"], "algorithmic_bytes_per_launch": round(d["bytes_per_launch"]),
            # the same launch against the memory roof: below the ridge (peak FLOP/s / peak B/s) the algorithmic bytes,
            # not the matrix pipe, set the floor of a launch
            "hbm_side": {"achieved": round(d["bytes_per_launch"] / (d["avg_launch_us"] * 1e-6) / 1e9, 1),
                         "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                         "frac": round(d["bytes_per_launch"] / (d["avg_launch_us"] * 1e-6) / 1e9 / PEAK_HBM_GBPS, 4),
                         "flop_per_byte": round(d["flop_per_launch"] / d["bytes_per_launch"], 1),
                         "ridge_flop_per_byte": round(PEAK_BF16_TFLOPS * 1e12 / (PEAK_HBM_GBPS * 1e9), 1)},
            "how": "whole-batch launches of every conv shape of one forward, in the forward's launch mix, back to back on "
                   "one stream between two HIP events; the timed forwards themselves run two half-batches on two "
                   "streams, where launches overlap and a per-launch duration is not a kernel rate"}
    other = {names[s]: {"achieved": round(v["achieved"], 2), "frac": round(v["achieved"] / PEAK_BF16_TFLOPS, 4),
                        "avg_launch_us": round(v["avg_launch_us"], 2), "launches_per_forward": v["launches_per_forward"],
                        "traffic": measured_traffic(symbols[s]),
                        "algorithmic_bytes_per_launch": round(v["bytes_per_launch"])} for s, v in out.items() if s != dom}
    per_shape = [{"kernel": sym, "shape": name, "us": round(sec * 1e6, 2), "tflops": round(fl / sec / 1e12, 1)}
                 for sym, name, _c, fl, sec, _b in rows]
    return roof, other, per_shape


def host_buffer_rate(net, B, H, W, dev, steps=20):
    """The image path with the batch starting and ending in pinned host memory: uint8 HWC host -> device, the
    generator with byte conversion fused at both ends, uint8 HWC device -> host.  A reported extra; the headline
    value keeps its inputs resident in HBM."""
    src = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(7)).pin_memory()
    dst = torch.empty((B, 4 * H, 4 * W, 3), dtype=torch.uint8).pin_memory()
    with torch.no_grad():
        for _ in range(3):
            dst.copy_(net.upscale_u8(src.to(dev, non_blocking=True)), non_blocking=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            dst.copy_(net.upscale_u8(src.to(dev, non_blocking=True)), non_blocking=True)
        torch.cuda.synchronize()
        sec = (time.perf_counter() - t0) / steps
    return {"value": round(B * 16 * H * W / sec / 1e6, 2), "unit": "HR Mpix/s", "ms_per_step": round(sec * 1e3, 3),
            "what": f"uint8 [N,H,W,3] batch of {B} from pinned host memory, x4 image back into pinned host memory "
                    f"({B * H * W * 3 / 1e6:.2f} MB in, {B * 16 * H * W * 3 / 1e6:.2f} MB out per step), one stream, no "
                    "copy/compute overlap between steps"}


def _cpu_description():
    """CPU model and physical-core count of the machine (from /proc/cpuinfo) beside the CPUs this process may use."""
    model, cores = "unknown", set()
    try:
        phys = core = None
        for ln in open("/proc/cpuinfo"):
            k, _, v = ln.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif not k and phys is not None and core is not None:
                cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = os.cpu_count() or 1
    return model, len(cores) or (os.cpu_count() or 1), usable


def cpu_baseline(warmup=3, timed=10):
    """The oracle (a port of the reference PyTorch forward, fp32) on the host cores, on BASELINE configs[0] as it stands:
    one batch of 4 LR patches of 3x64x64, weights seed 0, input seed 1234, no_grad; ``warmup`` untimed calls, the
    median of ``timed`` calls.  Threads: what torch was given on this box (OMP_NUM_THREADS is the box's CPU share — the
    machine's physical cores are not all this process's to use); the count actually used is reported as ``cores``."""
    from oracle import esrgan_oracle as O
    sd = O.make_generator_state(0)
    B, H, W = 4, 64, 64
    x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(1234))
    model, physical, usable = _cpu_description()
    cores = torch.get_num_threads()
    with torch.no_grad():
        for _ in range(warmup):
            O.generator_forward(sd, x)
        times = []
        for _ in range(timed):
            t0 = time.perf_counter()
            O.generator_forward(sd, x)
            times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    return {"value": round(B * 16 * H * W / med / 1e6, 4), "unit": "HR Mpix/s", "cores": cores, "kind": "port",
            "cpu_model": model, "machine_physical_cores": physical, "cpus_usable_by_this_process": usable,
            "sample": f"configs[0]: batch {B} of 3x{H}x{W} fp32 LR patches, median of {len(times)} forwards after "
                      f"{warmup} warm-up calls, {med:.3f} s each, torch CPU threads={cores}"}


def _rate(fn, steps, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def other_configs(net, dev, world, rank, dist):
    """The BASELINE configurations besides the headline one, as extras of the same JSON line:
    configs[0] (batch 4 of 64x64, the CPU case) on the GPU; configs[3] (a DIV2K-size 510x339 LR image -> 2040x1356)
    whole, as four / eight tiles with halo on one GPU, and sharded one tile per rank with the final
    gather (N > 1); configs[4] (CycleGAN GeneratorSR and the conditional VAE on the same conv kernels)."""
    import super_resolution_amd as S
    out = {}
    with torch.no_grad():
        if rank == 0:
            x0 = torch.rand(4, 3, 64, 64, device=dev)
            sec = _rate(lambda: net(x0), 20)
            out["configs[0] batch 4 of 3x64x64 on one GPU"] = {"value": round(4 * 16 * 64 * 64 / sec / 1e6, 2),
                                                               "unit": "HR Mpix/s", "ms_per_step": round(sec * 1e3, 3)}
        img = torch.rand(1, 3, 339, 510, generator=torch.Generator().manual_seed(99)).to(dev)
        hr_mpix = 16 * 339 * 510 / 1e6
        if rank == 0:
            sec = _rate(lambda: net(img), 5, 1)
            whole = net(img)
            out["configs[3] 510x339 whole image on one GPU"] = {"value": round(hr_mpix / sec, 2), "unit": "HR Mpix/s",
                                                                "ms_per_image": round(sec * 1e3, 2)}
            from super_resolution_amd import sharded
            for tiles in (4, 8):
                g = sharded.best_grid(339, 510, tiles, sharded.DEFAULT_HALO)
                wf = sharded.work_factor(339, 510, sharded.tile_edges(339, 510, g[0], g[1], sharded.DEFAULT_HALO))
                sec = _rate(lambda: S.tiled_forward(net, img, tiles), 5, 1)
                tiled = S.tiled_forward(net, img, tiles)
                out[f"configs[3] 510x339 as {tiles} tiles ({g[0]} x {g[1]}, halo {sharded.DEFAULT_HALO}, one batch) on one GPU"] = {
                    "value": round(hr_mpix / sec, 2), "unit": "HR Mpix/s", "ms_per_image": round(sec * 1e3, 2),
                    "work_factor": round(wf, 3), "max_abs_vs_whole": float((tiled - whole).abs().max())}
            sec = _rate(lambda: S.tiled_forward(net, img, 8, halo=16, grid=(8, 1), batched=False), 3, 1)
            out["configs[3] 510x339 as 8 row bands, halo 16, one after the other (row bands instead of a tile grid)"] = {
                "value": round(hr_mpix / sec, 2), "unit": "HR Mpix/s", "ms_per_image": round(sec * 1e3, 2)}
            del whole, tiled
        if world > 1:
            def sharded():
                return S.sharded_forward(net, img)
            for _ in range(2):
                sharded()
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                sharded()
            dist.barrier()
            torch.cuda.synchronize()
            sec = _max_over_ranks((time.perf_counter() - t0) / 5, dev, dist)
            out[f"configs[3] 510x339 sharded as one tile per rank over {world} GPUs (halo 8, gather on rank 0)"] = {
                "value": round(hr_mpix / sec, 2), "unit": "HR Mpix/s", "ms_per_image": round(sec * 1e3, 2),
                "scaling": "strong"}
        if rank == 0:
            gsr = S.cyclegan.GeneratorSR().to(dev).eval()
            xg = torch.rand(16, 3, 64, 64, device=dev) * 2 - 1
            sec = _rate(lambda: gsr(xg), 20)
            y = gsr(xg)
            out["configs[4] CycleGAN GeneratorSR (6 residual blocks), batch 16 of 3x64x64"] = {
                "value": round(16 * y.shape[2] * y.shape[3] / sec / 1e6, 2), "unit": "HR Mpix/s",
                "ms_per_step": round(sec * 1e3, 3), "out": list(y.shape)}
            vae = S.cvae.CVAE().to(dev).eval()
            xv = torch.rand(16, 3, 32, 32, device=dev)
            sec = _rate(lambda: vae(xv), 20)
            yv = vae(xv)[0]
            out["configs[4] conditional VAE forward (encode, reparameterise, decode), batch 16 of 3x32x32"] = {
                "value": round(16 / sec, 1), "unit": "images/s", "ms_per_step": round(sec * 1e3, 3), "out": list(yv.shape)}
    return out


def strong_scaling(net, dev, world, dist, B, H, W, steps):
    """The headline batch split over the ranks (fixed total work): every rank runs B / world patches."""
    if world == 1 or B % world:
        return None
    x = torch.rand(B // world, 3, H, W, device=dev)
    with torch.no_grad():
        for _ in range(3):
            net(x)
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            net(x)
        dist.barrier()
        torch.cuda.synchronize()
    el = _max_over_ranks(time.perf_counter() - t0, dev, dist)
    return {"value": round(B * 16 * H * W * steps / el / 1e6, 2), "unit": "HR Mpix/s", "ms_per_step": round(el / steps * 1e3, 3),
            "scaling": "strong", "global_batch": B, "batch_per_gpu": B // world}


def collective_rates(dev, world, dist, reps=10):
    """The mean of the generator's 66.8 MB flat gradient buffer over the ranks, both ways the library can run it: one
    RCCL all-reduce, and the direct reduce-scatter + all-gather (two all-to-all exchanges, ddp.direct_mean_)."""
    from super_resolution_amd import ddp
    n = 16_697_987
    buf = torch.randn(n, device=dev)
    out = {"floats": n, "world": world}
    for alg in ("all_reduce", "direct"):
        for _ in range(2):
            ddp.all_reduce_mean(buf, None, alg)
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            ddp.all_reduce_mean(buf, None, alg)
        torch.cuda.synchronize()
        sec = _max_over_ranks((time.perf_counter() - t0) / reps, dev, dist)
        out[alg] = {"ms": round(sec * 1e3, 3), "algorithmic_GBps_per_rank": round(2 * (world - 1) / world * 4 * n / sec / 1e9, 1)}
    return out


def _max_over_ranks(seconds, dev, dist):
    # RCCL reduces device tensors; the gloo rehearsal reduces on the host
    t = torch.tensor([seconds], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def _timed_steps(step, dev, dist, steps, warmup):
    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    el = time.perf_counter() - t0
    if dist is not None:
        el = _max_over_ranks(el, dev, dist)
    return el


def graphed_pixel_step(dev, lr, hr, steps):
    """The same pixel pre-training iteration captured once into a HIP graph (forward, loss, backward with its
    side-stream weight gradients, Adam) and replayed: the host issues one launch per step."""
    import super_resolution_amd as S
    try:
        torch.manual_seed(0)
        net = S.GeneratorRRDB(precision="bf16").to(dev).train()
        opt = torch.optim.Adam(net.parameters(), lr=1e-4, betas=(0.9, 0.999), capturable=True)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # warm-up off the default stream, as capture requires
            for _ in range(3):
                opt.zero_grad(set_to_none=True)
                torch.nn.functional.l1_loss(net(lr), hr).backward()
                opt.step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        opt.zero_grad(set_to_none=True)
        with torch.cuda.graph(graph):
            loss = torch.nn.functional.l1_loss(net(lr), hr)
            loss.backward()
            opt.step()
        before = net.conv_first.weight.detach().clone()
        graph.replay()
        torch.cuda.synchronize()
        first = float(loss.detach())
        t0 = time.perf_counter()
        for _ in range(steps):
            graph.replay()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        moved = not torch.equal(before, net.conv_first.weight.detach())
        ok = moved and float(loss.detach()) < first and bool(torch.isfinite(loss.detach()))      # replays train: the loss falls
        return {"value": round(16 * steps / el, 1), "unit": "patches/s", "ms_per_step": round(el / steps * 1e3, 2),
                "replays_update_the_weights_and_lower_the_loss": ok}
    except Exception as exc:
        try:
            torch.cuda.synchronize()
        except Exception:
            pass
        return {"error": f"{type(exc).__name__}: {str(exc)[:300]}"}


def train_kernel_roofline(dev, batches=(16, 32, 64), size=32):
    """The three kernels that make up four fifths of a training iteration, timed on their own at the training shape
    (LR patches of 32x32) for several per-GPU batch sizes: the one-launch dense block (forward form, as the recorded
    forward runs it), its data-gradient form (sign-gated by the saved activations) and the batched weight gradient of
    a dense block's five convs.  Each is launched back to back through the C ABI between two HIP events on torch's
    current stream (the stream the launches go to); FLOPs are 2*9*Cin*Cout per pixel and conv, the algorithmic bytes
    what each kernel must read and write once per pixel (weights, 479 KB per block, not counted)."""
    import super_resolution_amd as S
    from super_resolution_amd import ops
    nf, gc, units = 64, 32, 24
    torch.manual_seed(0)
    rdb = S.ResidualDenseBlock(nf, gc).to(dev)
    packed = rdb._packed(S.DTYPE_BF16)
    packs_t = rdb._packed_bwd()
    zero = torch.zeros(64, dtype=torch.float32, device=dev)
    flop_px = 2.0 * 9 * sum((nf + k * gc) * (gc if k < 4 else nf) for k in range(5))
    bytes_px = {"dense_block_band (forward)": 2 * nf + 4 * nf + 4 * 2 * gc + 2 * nf + 4 * nf,
                "dense_block_band_gated (data gradient)": 2 * nf + 4 * 2 * gc + 4 * 2 * gc + 4 * nf,
                "wgrad3x3_kernel_mb1 (five convs) + reduction": 2 * 2 * (nf + 4 * gc)}
    WGRAD6 = "wgrad3x3_kernel_mb1 + reduction, nine blocks per launch, per block"
    bytes_px[WGRAD6] = bytes_px["wgrad3x3_kernel_mb1 (five convs) + reduction"]
    out = {"shape": f"LR patches of {size}x{size}, one dense block (64 + 4 x 32 channels)", "flop_per_pixel": flop_px,
           "algorithmic_bytes_per_pixel": bytes_px, "peak_tflops": PEAK_BF16_TFLOPS, "peak_gbps": PEAK_HBM_GBPS,
           "per_batch": {}}

    def timed(launch, arg, reps=200):
        for _ in range(10):
            launch(arg)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            launch(arg)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / reps

    for B in batches:
        g = torch.Generator(device=dev).manual_seed(B)
        buf = torch.randn((B, size, units, size, 8), generator=g, device=dev).to(torch.bfloat16)
        nxt = torch.empty_like(buf)
        stack = torch.randn((B, size, units, size, 8), generator=g, device=dev).to(torch.bfloat16)
        trunk = torch.randn((B, size, nf // 4, size, 4), generator=g, device=dev)
        dxc = torch.empty_like(trunk)
        mask_off = [(nf + (3 - k) * gc) // 8 for k in range(4)]
        fwd = ops.dense_block_args(buf, [w for w, _ in packed], [b for _, b in packed], 0.2, True, trunk, 0.2, None, 0.0,
                                   nxt, trunk)
        gated = ops.dense_block_args(stack, packs_t, [zero] * 5, 1.0, True, None, 1.0, None, 0.0, None, dxc, mask=buf,
                                     mask_unit_off=mask_off, mask_slope=0.2)
        dws = [torch.empty((gc if k < 4 else nf, nf + k * gc, 3, 3), dtype=torch.float32, device=dev) for k in range(5)]
        dbs = [torch.empty((gc if k < 4 else nf,), dtype=torch.float32, device=dev) for k in range(5)]
        need = ops._lib.lib().sr355_dense_block_wgrad_workspace_bytes(nf, gc, B, size, size)
        wsb = torch.empty(need, dtype=torch.uint8, device=dev)
        wg = ops.dense_wgrad_args(buf, stack, nf, gc, 1.0, 0.2, dws, dbs, wsb)
        # nine blocks per launch, each with buffers of its own: the form the generator's backward runs (three RRDBs' weight
        # gradients behind the ninth block's data gradients); the figure is per block
        six_ws = torch.empty(9 * need, dtype=torch.uint8, device=dev)
        six_bufs = [(torch.randn((B, size, units, size, 8), generator=g, device=dev).to(torch.bfloat16),
                     torch.randn((B, size, units, size, 8), generator=g, device=dev).to(torch.bfloat16)) for _ in range(9)]
        six = [ops.dense_wgrad_args(a_, s_, nf, gc, 1.0, 0.2, dws, dbs, six_ws) for a_, s_ in six_bufs]
        px = B * size * size
        rows = {}
        for name, launch, arg, per in (("dense_block_band (forward)", ops.launch_dense_block, fwd, 1),
                                       ("dense_block_band_gated (data gradient)", ops.launch_dense_block, gated, 1),
                                       ("wgrad3x3_kernel_mb1 (five convs) + reduction", ops.launch_dense_wgrad, wg, 1),
                                       (WGRAD6, ops.launch_dense_wgrad_blocks, six, 9)):
            sec = timed(launch, arg) / per
            rows[name] = {"avg_launch_us": round(sec * 1e6, 2), "tflops": round(flop_px * px / sec / 1e12, 1),
                          "mfma_frac": round(flop_px * px / sec / 1e12 / PEAK_BF16_TFLOPS, 4),
                          "gbps": round(bytes_px[name] * px / sec / 1e9, 1),
                          "hbm_frac": round(bytes_px[name] * px / sec / 1e9 / PEAK_HBM_GBPS, 4)}
        out["per_batch"][str(B)] = rows
    return out


def train_rate(dev, world, dist, steps=20, warmup=5, graph=False):
    """Extra figures beside the headline, on 16 LR patches of 32x32 (128x128 HR) per GPU (BASELINE config 3 shapes):
    * the pixel pre-training iteration (esrgan/train.py:157-163: G forward, L1, backward, Adam);
    * the adversarial iteration (:196-223: D update on (hr, fake), then G update with L1 + 0.01 VGG-L1 + 1e-3 GAN loss),
      VGG19 with seeded random weights (throughput does not depend on the weights).
    With more than one rank the generator's gradients go through the bucketed all-reduce overlapped with its backward
    and the discriminator's through one all-reduce of its flat gradient buffer in front of its optimizer step."""
    import super_resolution_amd as S
    from super_resolution_amd.ddp import DistributedGenerator

    def note(msg):          # progress on stderr: the multi-rank path has collectives a stalled peer can block
        if world > 1:
            print(f"[bench rank {os.environ.get('RANK', '0')}] {msg}", file=sys.stderr, flush=True)

    torch.manual_seed(0)
    netG = S.GeneratorRRDB(precision="bf16").to(dev).train()
    modelG = DistributedGenerator(netG) if world > 1 else netG
    def adam(params, reduce_group=None):
        # the reference's optimizer (train.py:117-122) as one launch over a flat parameter buffer: the same update
        # (tests/test_gpu_training.py::test_flat_adam_is_torch_adam), without the per-tensor host work of torch's Adam
        return (S.FlatAdam(params, lr=1e-4, betas=(0.9, 0.999), reduce_group=reduce_group),
                "FlatAdam(1e-4, betas 0.9/0.999) = torch.optim.Adam's update in one launch")

    optG, opt_name = adam(netG.parameters())
    lr = torch.rand(16, 3, 32, 32, device=dev)
    hr = torch.rand(16, 3, 128, 128, device=dev)
    losses = {}

    def pixel():
        losses.update(S.pixel_step(modelG, optG, lr, hr))

    note("pixel iterations")
    el = _timed_steps(pixel, dev, dist, steps, warmup)
    note("pixel iterations done")
    assert bool(torch.isfinite(losses["loss_pix"]))
    flop = 3 * FLOP_PER_LR_PIXEL * 16 * 32 * 32          # backward = 2 forward-equivalents
    out = {"pixel_step": {"value": round(world * 16 * steps / el, 1), "unit": "patches/s",
                          "ms_per_step": round(el / steps * 1e3, 2), "generator_tflops": round(world * flop * steps / el / 1e12, 1)},
           "steps": steps, "warmup": warmup, "batch_per_gpu": 16, "lr_size": [32, 32], "hr_size": [128, 128],
           "optimizer": opt_name}
    if world == 1 and graph:
        out["pixel_step_hipgraph"] = graphed_pixel_step(dev, lr, hr, steps)
    if world == 1:
        # per-GPU batch sweep (--batch_size is a flag of the reference, esrgan/train.py:20): where the step saturates
        sweep = {}
        for bsz in (32, 64):
            try:
                lr_b, hr_b = torch.rand(bsz, 3, 32, 32, device=dev), torch.rand(bsz, 3, 128, 128, device=dev)
                el_b = _timed_steps(lambda: S.pixel_step(modelG, optG, lr_b, hr_b), dev, dist, steps, warmup)
                sweep[str(bsz)] = {"value": round(bsz * steps / el_b, 1), "unit": "patches/s",
                                   "ms_per_step": round(el_b / steps * 1e3, 2),
                                   "generator_tflops": round(3 * FLOP_PER_LR_PIXEL * bsz * 32 * 32 * steps / el_b / 1e12, 1)}
            except Exception as exc:
                sweep[str(bsz)] = {"error": f"{type(exc).__name__}: {str(exc)[:200]}"}
        out["pixel_step_batch_sweep"] = sweep
        try:
            out["roofline"] = train_kernel_roofline(dev)
        except Exception as exc:
            out["roofline"] = {"error": f"{type(exc).__name__}: {str(exc)[:300]}"}
    try:
        netD = S.Discriminator().to(dev).train()
        vgg = S.VGGFeatureExtractor(pretrained=False).to(dev)
        modelD = netD
        optD, _ = adam(netD.parameters(), dist.group.WORLD if world > 1 else None)

        def adversarial():
            losses.update(S.adversarial_step(modelG, modelD, vgg, optG, optD, lr, hr))

        note("adversarial iterations")
        el = _timed_steps(adversarial, dev, dist, steps, warmup)
        note("adversarial iterations done")
        assert bool(torch.isfinite(losses["loss_G"])) and bool(torch.isfinite(losses["loss_D"]))
        out["adversarial_step"] = {"value": round(world * 16 * steps / el, 1), "unit": "patches/s",
                                   "ms_per_step": round(el / steps * 1e3, 2),
                                   "scope": "D update + G update (one recorded G forward serving both, one G backward; the D "
                                            "update's two passes (real, fake) as one batched forward + backward with "
                                            "BatchNorm per half, a third D forward with its data-gradient-only pass for "
                                            "the G loss; the VGG19 features of (fake, hr) as one batched forward, one "
                                            "VGG19 data-gradient pass on the fake half), Adam on both"}
    except Exception as exc:
        out["adversarial_step"] = {"error": f"{type(exc).__name__}: {exc}"}
    return out


def launch_ranks(n, argv, env=None, python=None, script=None):
    """``bench.py --gpus N`` started plainly (no launcher): this process touches no GPU; it starts N fresh rank
    processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their environment, as
    torch.distributed.run would set them), relays rank 0's JSON line and returns the worst exit status."""
    import socket
    import subprocess
    base = dict(os.environ if env is None else env)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    base.setdefault("MASTER_ADDR", "127.0.0.1")
    base.setdefault("MASTER_PORT", str(port))
    base["WORLD_SIZE"] = base["LOCAL_WORLD_SIZE"] = str(n)
    procs = []
    for r in range(n):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([python or sys.executable, script or os.path.abspath(__file__), *argv], env=e,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    lines = [ln for ln in (out or "").splitlines() if ln.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    worst = max(codes, key=lambda c: (c != 0, abs(c)))
    return worst if worst >= 0 else 128 - worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--size", type=int, default=128, help="LR patch height = width")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the extra training-step figure")
    ap.add_argument("--no-extras", action="store_true", help="skip the other BASELINE configurations and strong scaling")
    ap.add_argument("--train-graph", action="store_true",
                    help="also replay the pixel iteration from a captured HIP graph (measured slower than eager issue: "
                         "about 3,000 graph nodes per step)")
    ap.add_argument("--hr-side", choices=("f16", "split"), default="f16",
                    help="arithmetic of the four HR-side convs in the timed (no-grad) forward: IEEE fp16 (default of the "
                         "module, 58.6 dB) or bf16 with a second weight image holding the rounding residue (53.9 dB)")
    ap.add_argument("--dense-skip-fp32", action="store_true",
                    help="keep an fp32 carrier for the dense-block skips too (58.6 dB instead of 55.6 dB, more epilogue bytes)")
    ap.add_argument("--streams", type=int, default=0,
                    help="batch parts the forward runs on concurrent HIP streams (0 = library default, 2); "
                         "1 serialises the kernels, which is what a per-kernel profile needs")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started without a launcher: become the launcher (before anything here has touched a GPU)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the product path has no CPU fallback)")
    # rehearsal of the multi-rank path on a one-GPU box: SR355_BENCH_REHEARSAL=1 puts every rank on device 0 and
    # uses gloo (RCCL refuses two ranks on one device); the numbers of such a run mean nothing
    rehearsal = world > 1 and os.environ.get("SR355_BENCH_REHEARSAL") == "1"
    if rehearsal:
        # ranks share one GPU here: the body's persistent launch needs all its workgroups resident, which two
        # processes on one device cannot both have
        os.environ["SR355_CHAIN"] = "0"
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    if args.streams > 0:
        os.environ["SR355_STREAMS"] = str(args.streams)      # read once, when the library first runs a forward
    import super_resolution_amd as S
    torch.manual_seed(0)
    net = S.GeneratorRRDB(precision="bf16", hr_f16=args.hr_side == "f16",
                          dense_skip_fp32=args.dense_skip_fp32).to(dev).eval()
    B, H, W = args.batch, args.size, args.size
    x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(1234 + rank)).to(dev)

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(args.warmup):
            out = net(x)
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = net(x)
        sync()
        elapsed = time.perf_counter() - t0
    assert out.shape == (B, 3, 4 * H, 4 * W) and bool(torch.isfinite(out).all())
    if dist is not None:
        elapsed = _max_over_ranks(elapsed, dev, dist)

    ms = elapsed / args.steps * 1e3
    hr_mpix = world * B * 16 * H * W * args.steps / elapsed / 1e6
    line = {
        "metric": "RRDBNet x4 HR Mpix/s (infer)", "value": round(hr_mpix, 2), "unit": "HR Mpix/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"configs[1]: RRDBNet x4 (nf=64, 23 RRDB) bf16 inference, batch {B} of 3x{H}x{W} "
                               f"LR patches per GPU -> 3x{4 * H}x{4 * W}, inputs resident in HBM",
                   "batch_per_gpu": B, "lr_size": [H, W], "parallelism": f"patch-sharded x{world}, no collective",
                   "streams_per_gpu": "1 (body as one persistent launch)" if os.environ.get("SR355_CHAIN", "1") != "0"
                                      else (args.streams if args.streams > 0 else 2),
                   "hr_side": "fp16 (4 of 351 convs, 8 % of the FLOPs; the other 347 in bf16)" if args.hr_side == "f16"
                              else "bf16 + weight rounding residue",
                   "skip_carriers": "fp32 for every skip" if args.dense_skip_fp32
                                    else "fp32 for the RRDB and global skips, bf16 activations for the dense-block skips"},
        "generator_tflops": round(world * B * H * W * FLOP_PER_LR_PIXEL * args.steps / elapsed / 1e12, 1),
    }
    if rank == 0:
        if not args.no_roofline:
            with torch.no_grad():
                rows = measure_kernels(net, B, H, W)
                roof, other, per_shape = roofline_from(rows)
            chain = chain_roofline(net, x, rows)
            if chain is not None:      # the body runs as one launch: that kernel dominates; the per-conv kernels it
                other = dict(other)    # replaces stay listed (they run the HR side and smaller workloads)
                other[roof.pop("kernel") + " [one launch per conv: SR355_CHAIN=0]"] = {
                    k: roof[k] for k in ("achieved", "frac", "avg_launch_us", "launches_per_forward", "traffic",
                                         "algorithmic_bytes_per_launch")}
                roof = chain
            line["roofline"] = roof
            line["roofline_other_kernels"] = other
            line["kernel_shapes"] = per_shape
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline()
        if world == 1 and not args.no_roofline:
            try:
                line["pcie_inclusive"] = host_buffer_rate(net, B, H, W, dev)
            except Exception as exc:      # an extra: the headline number stands on its own
                line["pcie_inclusive"] = {"error": f"{type(exc).__name__}: {exc}"}
    extra = {}       # filled by extras() (possibly on a worker thread), merged into the line only once it has finished

    def extras():
        """Everything beside the headline: the other BASELINE configurations, strong scaling, the training iterations."""
        if not args.no_extras:
            try:
                if world > 1:
                    print(f"[bench rank {rank}] strong scaling", file=sys.stderr, flush=True)
                    extra["strong_scaling"] = strong_scaling(net, dev, world, dist, B, H, W, max(5, args.steps // 2))
                    print(f"[bench rank {rank}] gradient mean, both algorithms", file=sys.stderr, flush=True)
                    extra["gradient_mean_over_ranks"] = collective_rates(dev, world, dist)
                    print(f"[bench rank {rank}] other configurations", file=sys.stderr, flush=True)
                others = other_configs(net, dev, world, rank, dist)
                if rank == 0:
                    extra["other_configs"] = others
            except Exception as exc:          # extras: the headline number stands on its own
                extra["other_configs"] = {"error": f"{type(exc).__name__}: {str(exc)[:300]}"}
        if not args.no_train:
            try:
                extra["train"] = train_rate(dev, world, dist, graph=args.train_graph and world == 1)
            except Exception as exc:
                extra["train"] = {"error": f"{type(exc).__name__}: {exc}"}

    hung = False
    del out
    if world == 1:
        extras()
    else:
        # collectives inside: a rank that fails or stalls must not take the headline with it, so with more than one
        # rank the extras run on a worker thread with a deadline
        import threading

        def work():
            torch.cuda.set_device(dev)        # the current device is per thread
            extras()

        th = threading.Thread(target=work, daemon=True)
        th.start()
        th.join(timeout=240.0)
        hung = th.is_alive()
        if hung:      # the worker still writes to `extra`: it is not read
            line["extras_error"] = "the extras (other configurations, strong scaling, training) did not finish within 240 s"
    if not hung:
        line.update(extra)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if hung:
        os._exit(0)        # a stalled collective cannot be cancelled; the line is out
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
