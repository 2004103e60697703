"""TEST INFRASTRUCTURE: builds the recording HIP stand-in (hipmock.cpp), runs drive.py scenarios under it in a subprocess and
parses / digests the traces.  Used by tests/test_launch_trace.py and tests/golden/make_launch_traces.py."""
import hashlib
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
PRODUCT_LIB = os.path.join(ROOT, "bndm_amd", "libbndm_hip.so")
SCENARIOS = ("c2", "c2bf16", "c3", "c4", "c5", "cond", "f32", "noise", "b500", "b200")
DEV_LO, DEV_HI = 0x200000000000, 0x200000000000 + (1 << 40)          # hipmock.cpp: kBase, kSpan


def build_mock(outdir):
    """-> directory holding libamdhip64.so.7 (the stand-in)"""
    os.makedirs(outdir, exist_ok=True)
    so = os.path.join(outdir, "libamdhip64.so.7")
    src = os.path.join(HERE, "hipmock.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                        "-Wl,-soname,libamdhip64.so.7", f"-Wl,--version-script={os.path.join(HERE, 'hipmock.map')}", "-o", so, src],
                       check=True)
    return outdir


def kernargs_file(lib, outdir):
    dst = os.path.join(outdir, "kernargs_" + hashlib.sha256(open(lib, "rb").read()).hexdigest()[:12] + ".txt")
    if not os.path.exists(dst):
        subprocess.run([sys.executable, os.path.join(HERE, "kernargs.py"), lib, dst], check=True, capture_output=True)
    return dst


_MASKS = {}


def read_masks(lib, outdir):
    """per kernel, which bytes of each explicit argument the machine code reads (kernargs.read_masks), cached next to the argument sizes"""
    import json
    key = os.path.abspath(lib)
    if key not in _MASKS:
        path = kernargs_file(lib, outdir)[:-4] + "_readmasks.json"
        if not os.path.exists(path):
            from tests.hipmock import kernargs
            m = kernargs.read_masks(lib)
            with open(path, "w") as f:
                json.dump({k: None if v is None else [x.hex() for x in v] for k, v in m.items()}, f)
        raw = json.load(open(path))
        _MASKS[key] = {k: None if v is None else [bytes.fromhex(x) for x in v] for k, v in raw.items()}
    return _MASKS[key]


def _mask_unread(ln, masks):
    """zero the argument bytes a kernel never reads (struct padding, unused fields): they are whatever the host stack held"""
    if not ln.startswith("launch ") or ln.endswith("args=?"):
        return ln
    sym = ln.split(" ", 2)[1]
    mk = masks.get(sym)
    if mk is None:
        return ln
    head, args = ln.rsplit("args=", 1)
    parts = args.split("|")
    if len(parts) != len(mk):
        return ln
    out = []
    for a, m in zip(parts, mk):
        b = bytearray(bytes.fromhex(a))
        if len(b) == len(m):
            for i, keep in enumerate(m):
                if not keep:
                    b[i] = 0
        out.append(b.hex())
    return head + "args=" + "|".join(out)


def run_scenario(lib, name, outdir, lanes=1, flags=0):
    """-> list of trace lines (kernel-argument bytes the kernels never read are zeroed: see kernargs.read_masks)"""
    mock = build_mock(outdir)
    trace = os.path.join(outdir, f"trace_{os.path.basename(lib)}_{name}_{lanes}_{flags}.txt")
    if os.path.exists(trace):
        os.remove(trace)
    env = dict(os.environ, LD_LIBRARY_PATH=mock + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""), HIPMOCK_TRACE=trace,
               HIPMOCK_KERNARGS=kernargs_file(lib, outdir))
    r = subprocess.run([sys.executable, os.path.join(HERE, "drive.py"), lib, name, str(lanes), str(flags)], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, f"drive.py {name} failed:\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
    masks = read_masks(lib, outdir)
    return [_mask_unread(_mask_padding(ln), masks) for ln in open(trace).read().splitlines()]


def run_script(script, lib, outdir, *args, env=None, mockdir=None, timeout=900):
    """runs tests/hipmock/<script> <lib> <args> under the stand-in -> its stdout (`env`: extra environment of that process)"""
    mock = build_mock(mockdir or outdir)
    os.makedirs(outdir, exist_ok=True)
    trace = os.path.join(outdir, f"trace_{script}_{os.path.basename(lib)}.txt")
    if os.path.exists(trace):
        os.remove(trace)
    env = dict(os.environ, **(env or {}), LD_LIBRARY_PATH=mock + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""), HIPMOCK_TRACE=trace,
               HIPMOCK_KERNARGS=kernargs_file(lib, mockdir or outdir))
    r = subprocess.run([sys.executable, os.path.join(HERE, script), lib, *map(str, args)], env=env, capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0, f"{script} failed:\n{r.stdout[-3000:]}\n{r.stderr[-3000:]}"
    return r.stdout


def _mask_padding(ln):
    """struct padding passed by value is whatever the host stack held: ZSrc (bluenoise.hip: pointer + 3 ints = 20 of 24 bytes) -- kept for
    bluenoise_finish, one of the kernels whose use of the kernarg pointer _mask_unread's analysis does not follow"""
    if ln.startswith("launch ") and "4ZSrcE" in ln:
        head, args = ln.rsplit("args=", 1)
        a = args.split("|")
        assert len(a[1]) == 48
        a[1] = a[1][:40] + "00000000"
        return head + "args=" + "|".join(a)
    return ln


_SHORT = re.compile(r"^_ZN(?:4bndm)?(?:12_GLOBAL__N_1)?(\d+)")


def short_name(sym):
    """kernel family from the mangled symbol: _ZN4bndm12_GLOBAL__N_18conv_t32I... -> conv_t32"""
    m = _SHORT.match(sym)
    if not m:
        return sym
    n = int(m.group(1))
    return sym[m.end():m.end() + n]


def parse_launch(line):
    """'launch <sym> g=.. b=.. lds=.. st=.. args=a|b|c' -> dict"""
    p = line.split(" ")
    d = dict(sym=p[1], name=short_name(p[1]))
    for kv in p[2:]:
        k, v = kv.split("=", 1)
        d[k] = v
    d["args"] = [] if d["args"] == "?" else [bytes.fromhex(a) for a in d["args"].split("|")]
    return d


def stages(lines):
    """[(mark, [lines])] -- the trace cut at the '== <mark>' lines ('op ...' marks stay inside their stage)"""
    out = [("create", [])]
    for ln in lines:
        if ln.startswith("== ") and not ln.startswith("== op "):
            out.append((ln[3:], []))
        else:
            out[-1][1].append(ln)
    return out


def digest(lines):
    """compact, diff-able record of a trace: per stage the sha256 of its text, the call counts, and one short record per launch"""
    out = []
    for mark, body in stages(lines):
        rec = {"stage": mark, "sha256": hashlib.sha256("\n".join(body).encode()).hexdigest(),
               "calls": {}, "launches": []}
        for ln in body:
            kind = ln.split(" ", 1)[0]
            rec["calls"][kind] = rec["calls"].get(kind, 0) + 1
            if kind == "launch":
                assert not ln.endswith("args=?"), f"a launch whose kernel is missing from the kernel-argument table is not compared: {ln[:120]}"
                d = parse_launch(ln)
                rec["launches"].append(f"{d['name']} g={d['g']} b={d['b']} lds={d['lds']} " + hashlib.sha256(ln.encode()).hexdigest()[:8])
            elif kind == "==":
                rec.setdefault("ops", []).append(ln[3:])
        out.append(rec)
    return out


def check_pointers(lines):
    """Every 8-byte-aligned kernel-argument word that points into the stand-in's device address space must lie inside a LIVE
    allocation (one-past-the-end allowed), and no copy may have gone out of range.  -> number of pointers checked"""
    live = {}
    n = 0
    for i, ln in enumerate(lines):
        if ln.startswith("malloc "):
            _, p, sz = ln.split()
            live[int(p, 16)] = int(sz)
        elif ln.startswith("free "):
            assert "INVALID" not in ln, f"line {i}: {ln}"
            live.pop(int(ln.split()[1], 16), None)
        elif "OUT-OF-RANGE" in ln:
            raise AssertionError(f"line {i}: {ln}")
        elif ln.startswith("launch "):
            d = parse_launch(ln)
            bases = sorted(live)
            import bisect
            for a in d["args"]:
                for o in range(0, len(a) - 7, 8):
                    v = int.from_bytes(a[o:o + 8], "little")
                    if DEV_LO <= v < DEV_HI:
                        j = bisect.bisect_right(bases, v) - 1
                        assert j >= 0 and v <= bases[j] + max(live[bases[j]], 1), \
                            f"line {i}: {d['name']} argument word {v:#x} is outside every live allocation"
                        n += 1
    return n
