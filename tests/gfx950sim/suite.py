"""TEST INFRASTRUCTURE: the configurations run on the instruction-level simulator, and the runner that executes one of them
(a library's machine code through tests/hipmock/exec_forward.py with EXEC_SIM=1, or tests/gfx950sim/cases.py) and compares the
result with oracle/ -- used by tests/test_gfx950sim.py (fast subset), tools/sim_suite.sh (everything, log under profiles/) and
tools/sim_candidates.sh (candidate libraries).

    python -m tests.gfx950sim.suite [--lib lib.so] [--procs 8] [--work dir] name [name ...] | all | fast
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _subst(t):
    """run the 4-wave conv_t32 variants (what the library picks at >= 448 workgroups, i.e. batch >= 28 at 64x64) on the
    recorded arguments of the 8-wave launches: same argument list, the variant's block size and LDS size"""
    return (f"conv_t32I{t}Li16ELi0ELi8ELi128ELi1E=>conv_t32I{t}Li16ELi0ELi4ELi128ELi0E:256:81920;"
            f"conv_t32I{t}Li8ELi0ELi8ELi128ELi1E=>conv_t32I{t}Li8ELi0ELi4ELi128ELi0E:256:63488")


S4, S4B = _subst("DF16_"), _subst("DF16b")

# name -> (kind, case, batch, env, bar on rel-L2 / max-abs, what it covers)
CONFIGS = {
    # ---- fast subset (default CPU suite) ------------------------------------------------------------------------------
    "steps": ("cases", "steps", None, {}, 0.0, "iadb_step x2, ddim_step, export_u8 x2, iadb_train_targets: bit-exact"),
    "noise_small64": ("cases", "noise:small64", None, {}, 1e-4, "bluenoise_small<W16> + finish, B=2 64 px"),
    "lat_t32x4": ("unet", "lat256", 1, {"GFX950SIM_SUBST": S4}, 2e-3,
                  "latent celeba_res256 layout (128,256,256) at 32 px: conv_t32 4-wave TH=16 / TH=8 (substituted), conv_s incl. "
                  "64-token attention, igemm downsampler, head"),
    "deep32_t32x4": ("unet", "deep32", 1, {"GFX950SIM_SUBST": S4, "EXEC_MAX_BATCH": "64", "GFX950SIM_STATS": "1"}, 2e-3,
                     "five levels 32 .. 2 px (128,128,256,256,512), 4x4 attention, handle sized for batch 64 (the benchmark's tile variants): conv_t32 "
                     "4-wave TH=16 -- the DOMINANT kernel -- and TH=8 (substituted), conv_s<TM=128> / <TM=64> at 8x8 / 4x4 / "
                     "2x2 incl. stride-2, nearest-2x, shortcut and q|k|v + attention launches, gn_small, igemm downsamplers, conv_in, head"),
    # ---- full suite (tools/sim_suite.sh; RUN_SIM_SLOW=1) ------------------------------------------------------------------
    "c2": ("unet", "c2", 1, {"EXEC_MAX_BATCH": "64"}, 2e-3, "cat_res64 3->6, the handle bench.py builds (conv_s<TM=128>), 8-wave conv_t32"),
    "c2_t32x4": ("unet", "c2", 1, {"EXEC_MAX_BATCH": "64", "GFX950SIM_SUBST": S4}, 2e-3, "... with the DOMINANT 4-wave conv_t32<TH=16> / <TH=8>"),
    "c2_t32x4_rev": ("unet", "c2", 1, {"EXEC_MAX_BATCH": "64", "GFX950SIM_SUBST": S4, "GFX950SIM_ORDER": "1"}, 2e-3, "... waves in reverse order"),
    "c2_b2_random": ("unet", "c2", 2, {"GFX950SIM_ORDER": "5"}, 2e-3, "B=2 (conv_s<TM=64>), waves in random order"),
    "c2_bf16_t32x4": ("unet", "c2bf16", 1, {"EXEC_MAX_BATCH": "64", "GFX950SIM_SUBST": S4B}, 1e-2, "bf16 storage / MFMA inputs"),
    "c2_bf16": ("unet", "c2bf16", 1, {"EXEC_MAX_BATCH": "64"}, 1e-2, "bf16 with the 8-wave conv_t32 variants (what a batch-1 call launches)"),
    "c3_ddim_loop": ("unet", "c3loop", 1, {"EXEC_MAX_BATCH": "64"}, 3e-3, "church_res64 3->3: two steps of the in-engine DDIM loop"),
    "c2_iadb_loop": ("unet", "c2loop", 1, {"EXEC_MAX_BATCH": "64"}, 2e-3, "two steps of the in-engine IADB loop with snapshots"),
    "c4": ("unet", "c4", 1, {"EXEC_MAX_BATCH": "32", "GFX950SIM_SUBST": S4}, 2e-3, "celeba_res128 3->6 (7 levels)"),
    "c4_b1_handle": ("unet", "c4", 1, {"EXEC_MAX_BATCH": "1"}, 2e-3, "celeba_res128 on a batch-1 handle: 128-pixel tiles everywhere (128 GroupNorm slabs at 128x128)"),
    "c5": ("unet", "c5", 1, {"EXEC_MAX_BATCH": "8"}, 2e-3, "latent 4->8 at c5's per-GPU batch handle"),
    "cond": ("unet", "cond", 1, {}, 2e-3, "conditional (super-resolution) sampler, two steps"),
    "lat256": ("unet", "lat256", 2, {}, 2e-3, "latent celeba_res256 layout, ragged batch"),
    "w64": ("unet", "w64", 1, {}, 2e-3, "first-level width 64 (conv_in swizzle fix)"),
    "w256": ("unet", "w256", 1, {}, 2e-3, "first-level width 256"),
    "w64x4": ("unet", "w64x4", 2, {}, 2e-3, "(64, 64, 128, 256) with 8x8 attention, B=2: test_gpu_first_level_widths.py's second layout"),
    "bottom1x1": ("unet", "bottom1x1", 2, {}, 2e-3, "1x1 bottom level: deferred split-K in front of a conv_s upsampler"),
    "no_tail": ("unet", "c2", 1, {"BNDM_NO_TAIL": "1"}, 2e-3, "fallback: <= 8x8 levels on conv_igemm + gn_small"),
    "no_fused": ("unet", "c2", 1, {"BNDM_NO_FUSED": "1"}, 2e-3, "fallback: no conv_t32 (igemm + materialised GroupNorm everywhere)"),
    "no_gn_small": ("unet", "lat256", 1, {"BNDM_NO_TAIL": "1", "BNDM_NO_GN_SMALL": "1"}, 2e-3, "fallback variant: no gn_small (gn_stats + finalize + apply at <= 8x8)"),
    "no_defer": ("unet", "lat256", 1, {"BNDM_NO_TAIL": "1", "BNDM_NO_DEFER": "1"}, 2e-3, "fallback variant: split-K sums by splitk_reduce instead of the consumer"),
    "f32mode": ("unet", "lat256f32", 1, {}, 1e-4, "fp32-compute verification mode (plain FMA kernels: the latent 32-px layout, c2 would be ~10^9 wave-instructions)"),
    "vae16": ("unet", "vae16", 1, {}, 5e-3, "AutoencoderKL decoder, full layout, 16x16 latent"),
    # ---- kernel variants no benchmark configuration launches (coverage: every reachable kernel of the library runs at least once) --------
    "igemm_256x128": ("unet", "big128", 12, {"BNDM_NO_FUSED": "1"}, 2e-3, "conv_igemm's 256x128 3-stage tile (plain + nearest-2x source): B=12 at 64x64x128 = 192 tiles"),
    "igemm_256x128_bf16": ("unet", "big128bf16", 12, {"BNDM_NO_FUSED": "1"}, 1e-2, "... bf16"),
    "generic_bf16": ("unet", "w64bf16", 1, {"BNDM_NO_FUSED": "1", "BNDM_NO_TAIL": "1"}, 1e-2, "bf16 instances of the fallback kernels: igemm tiles incl. the 128x32 head, gn_apply, attention_kernel"),
    "lat_bf16_t32x4": ("unet", "lat256bf16", 1, {"GFX950SIM_SUBST": S4B}, 1e-2, "bf16 conv_t32<TH=8> / 4-wave variants incl. the 32-channel head at 32 px"),
    "vae16_bf16": ("unet", "vae16bf16", 1, {}, 4e-2, "AutoencoderKL decoder in bf16 (softmax_rows<bf16>; no reference tolerance exists for this mode: the bar is 8x the f16 one)"),
    "f32mode_loop": ("unet", "loop16f32", 1, {}, 1e-4, "two steps of the in-engine IADB loop in fp32 mode (fill_f32_kernel)"),
    "noise_small32col": ("cases", "noise:small32col", None, {}, 1e-4, "bluenoise_small<W32>, 32-px crop, GBN"),
    "noise_gemm128": ("cases", "noise:gemm128", None, {}, 1e-4, "bluenoise_gemm, 128 px tile permutation + scrambled wn, a shard"),
    "noise_dense64": ("cases", "noise:dense64", None, {}, 1e-4, "l_dense = 1"),
    "noise_gemm64_b22": ("cases", "noise:gemm64_b22", None, {}, 1e-4, "bluenoise_gemm<NT=2>: 66 columns"),
    "noise_gemm64_b64": ("cases", "noise:gemm64_b64", None, {}, 1e-4, "bluenoise_gemm<NT=3>: 192 columns -- c2's own noise call (B = 64, 64 px)"),
}
FAST = ("steps", "noise_small64", "deep32_t32x4")


def expected(case, sd, cfg, out_dir):
    """what oracle/ computes for an exec_forward.py case -> (got, want) tensors"""
    import numpy as np
    import torch
    from oracle import unet_oracle as UO
    from tests.hipmock.exec_forward import CASES, DA, DDIM, DG, SIM_CASES, T_IN
    cin, cout, res, layout, B, mode = {**CASES, **SIM_CASES}[case]
    load = lambda what: torch.from_numpy(np.load(os.path.join(out_dir, f"exec_{case}_{what}.npy")))
    x = load("x")
    if mode == "vae":
        from oracle import vae_oracle as VO
        want = VO.decode(sd, cfg, x)
    elif mode == "forward":
        want = UO.forward(sd, cfg, x, load("t"))
    elif mode in ("iadb", "cond"):
        extra = load("extra") if mode == "cond" else None
        snaps = []
        for s in range(2):                                   # utils.py:196-226 / iadb_bn.py:384-438 with explicit tables
            d = UO.forward(sd, cfg, x if extra is None else torch.cat([x, extra], 1), T_IN[s])
            x = x + DA[s] * d[:, :x.shape[1]]
            if cout == 2 * x.shape[1]:
                x = x + DG[s] * d[:, x.shape[1]:]
            snaps.append(x)
        want = torch.stack(snaps)
    else:
        for s in range(2):                                   # ddim_diffusers.py:674-681
            t, sat, s1at, sap, s1ap = DDIM[5 * s:5 * s + 5]
            eps = UO.forward(sd, cfg, x, t)
            x0 = ((x - s1at * eps) / sat).clamp(-1.0, 1.0)
            x = sap * x0 + s1ap * eps
        want = x
    return load("out"), want


def run_config(name, lib=None, work=None, procs=8):
    """-> dict(name, ok, value, bar, hazards, launches, wave_instructions, seconds, detail)"""
    import subprocess
    from tests.hipmock import harness as H
    kind, case, batch, env, bar, what = CONFIGS[name]
    lib = os.path.abspath(lib or H.PRODUCT_LIB)
    work = work or os.path.join("/tmp", "gfx950sim_work")
    os.makedirs(work, exist_ok=True)
    H.build_mock(work)
    t0 = time.time()
    env = dict(env, GFX950SIM_PROCS=str(procs), OMP_NUM_THREADS="4", MKL_NUM_THREADS="4")
    cov = os.path.join(work, f"coverage_{name}.txt")
    if os.path.exists(cov):
        os.remove(cov)
    env["GFX950SIM_COVERAGE"] = cov

    def kernels_run():
        out = {}
        if os.path.exists(cov):
            for ln in open(cov):
                k, ni, hz = ln.split()
                e = out.setdefault(k, [0, 0])
                e[0] += 1
                e[1] += int(ni)
        return out
    if kind == "cases":
        e = dict(os.environ, **env, LD_LIBRARY_PATH=work + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""),
                 HIPMOCK_TRACE=os.path.join(work, f"trace_cases_{name}.txt"), HIPMOCK_KERNARGS=H.kernargs_file(lib, work))
        if os.path.exists(e["HIPMOCK_TRACE"]):
            os.remove(e["HIPMOCK_TRACE"])
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "gfx950sim", "cases.py"), lib, case], env=e, capture_output=True, text=True,
                           timeout=3000)
        line = next((ln for ln in r.stdout.splitlines() if ln.startswith("{")), None)
        if line is None:
            return dict(name=name, ok=False, detail=(r.stdout + r.stderr)[-3000:], seconds=time.time() - t0, what=what)
        j = json.loads(line)
        vals = [v for k, v in j.items() if isinstance(v, float) and k != "seconds"]
        return dict(name=name, ok=bool(j["ok"]) and r.returncode == 0, value=max(vals) if vals else 0.0, bar=bar, hazards=len(j["hazards"]),
                    launches=j["launches"], wave_instructions=j["wave_instructions"], seconds=round(time.time() - t0, 1), detail=j, what=what,
                    kernels=kernels_run())
    from tests import test_launch_trace as T
    import torch
    torch.set_num_threads(4)
    cfg, key, make = T._case_network(case)
    sd, wfile = T.oracle_weights(work, key, make)
    out_dir = os.path.join(work, name)
    env["EXEC_SIM"] = "1"
    if batch is not None:
        env["EXEC_BATCH"] = os.environ.get("SIM_BATCH", str(batch))          # (SIM_BATCH: another batch for the same configuration)
    try:
        out = H.run_script("exec_forward.py", lib, out_dir, out_dir, case, wfile, env=env, mockdir=work, timeout=3000)
    except (AssertionError, subprocess.TimeoutExpired) as e:
        return dict(name=name, ok=False, detail=str(e)[-3000:], seconds=round(time.time() - t0, 1), what=what)
    got, want = expected(case, sd, cfg, out_dir)
    rel = float((got - want).double().norm() / want.double().norm())
    import re
    m = re.search(r"OK simulated (\d+) launches; output rms [0-9.]+; hazards (\d+)", out)
    ninst = sum(int(x) for x in re.findall(r"launches\s+(\d+) wave-instructions", out))
    hz = int(m.group(2)) if m else -1
    import hashlib
    return dict(name=name, ok=bool(m) and hz == 0 and rel <= bar, value=rel, bar=bar, hazards=hz, launches=int(m.group(1)) if m else 0,
                wave_instructions=ninst, seconds=round(time.time() - t0, 1), detail=out[-1500:], what=what,
                out_hash=hashlib.sha256(got.numpy().tobytes()).hexdigest()[:16],      # equal hashes = bit-identical results
                kernels=kernels_run())


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib")
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--work")
    ap.add_argument("--coverage", help="write: every kernel of the library and the configurations of this run that executed it")
    ap.add_argument("names", nargs="+")
    a = ap.parse_args()
    names = list(CONFIGS) if a.names == ["all"] else list(FAST) if a.names == ["fast"] else a.names
    import hashlib
    from tests.hipmock import harness as H
    lib = os.path.abspath(a.lib or H.PRODUCT_LIB)
    print(f"library: {hashlib.sha256(open(lib, 'rb').read()).hexdigest()}  {lib}", flush=True)
    bad = 0
    ran = {}
    for n in names:
        r = run_config(n, lib, a.work, a.procs)
        if r["ok"]:
            for k, (nl, ni) in r.get("kernels", {}).items():
                ran.setdefault(k, []).append((n, nl, ni))
        v = r.get("value")
        print(f"{'PASS' if r['ok'] else 'FAIL'}  {n:18s} value {v if v is None else format(v, '.3e')} (bar {r.get('bar')})  hazards {r.get('hazards')}  "
              f"{r.get('launches', 0)} launches  {r.get('wave_instructions', 0)} wave-instructions  {r['seconds']} s  out {r.get('out_hash', '-')}   -- {r['what']}", flush=True)
        if not r["ok"]:
            bad += 1
            print("      " + str(r.get("detail"))[-2500:].replace("\n", "\n      "), flush=True)
    if a.coverage:
        write_coverage(a.coverage, lib, ran, names)
    sys.exit(1 if bad else 0)


# instantiated but not launchable through the library's host code (read off unet_kernels.hip::launch_conv_t / unet_engine.hip)
UNREACHABLE = {
    "Li4ELi1ELi1ELi1ELi0ELi4E": "conv_igemm 128x32 tile + NHWC16 epilogue: launch_conv_t instantiates the pair, the engine uses TILE_128x32 only for the NCHW32 head",
    "Li4ELi1ELi1ELi1ELi2ELi4ELb1E": "table-driven 128x32 head: the head's ConvArgs carries no step table, launch_conv_cfg always picks the generic walk",
    "Li4ELi2ELi2ELi2ELi1ELi3E": "conv_igemm 256x128 tile + split-K fp32 epilogue: plan_conv splits K only below 192 tiles, where it has already picked 128x128",
    "gn_finalize_kernel": "superseded by gn_finalize2_kernel: launch_gn_finalize has no caller",
}


def coverage_from_work(work, names):
    """{kernel: [(configuration, launches, wave-instructions)]} from the coverage_<name>.txt files a run left in its work directory"""
    ran = {}
    for n in names:
        per = {}
        for ln in open(os.path.join(work, f"coverage_{n}.txt")):
            k, ni, hz = ln.split()
            e = per.setdefault(k, [0, 0])
            e[0] += 1
            e[1] += int(ni)
        for k, (nl, ni) in per.items():
            ran.setdefault(k, []).append((n, nl, ni))
    return ran


def write_coverage(path, lib, ran, names):
    """every kernel in the library's gfx950 code objects x the PASSING configurations that executed it (hazard-free, result within the bar)"""
    import hashlib
    import re
    from tests.gfx950sim import loader
    ks = loader.load_library(lib)
    short = lambda k: re.sub(r"^_ZN4bndm12_GLOBAL__N_1\d+|^_ZN12_GLOBAL__N_1\d+", "", k)
    with open(path, "w") as f:
        f.write(f"# kernels of {lib} (sha256 {hashlib.sha256(open(lib, 'rb').read()).hexdigest()}) executed on the instruction-level simulator\n")
        f.write(f"# by the PASSING configurations of this run ({len(names)}: {' '.join(names)})\n")
        f.write(f"# {sum(1 for k in ks if k in ran)} of {len(ks)} kernels executed\n")
        f.write(f"# {'kernel':96s} launches  wave-instructions  configurations\n")
        for k in sorted(ks, key=short):
            e = ran.get(k)
            if e:
                f.write(f"{short(k)[:98]:98s} {sum(x[1] for x in e):8d} {sum(x[2] for x in e):18d}  {' '.join(x[0] for x in e)}\n")
        never = [k for k in sorted(ks, key=short) if k not in ran]
        dead = {k: next((why for sub, why in UNREACHABLE.items() if sub in k), None) for k in never}
        f.write(f"# not executed: {len(never)} ({sum(1 for v in dead.values() if v)} of them cannot be launched through the library's host code)\n")
        for k in never:
            f.write(f"{'UNREACHABLE' if dead[k] else 'NEVER'} {short(k)}" + (f"   -- {dead[k]}" if dead[k] else "") + "\n")


if __name__ == "__main__":
    main()
