"""TEST INFRASTRUCTURE: launches of the gfx950 simulator -- kernarg segment, initial register state, the wave scheduler of a
workgroup, and the hook that makes the recording HIP stand-in (tests/hipmock) EXECUTE the kernels it records.

    sim = Simulator(lib)            # parses the library's code objects (cached)
    sim.install()                   # under the stand-in: every hipLaunchKernel of this process now runs here, in order
"""
import ctypes as C
import os
import time

import numpy as np

from . import loader
from .core import BARRIER, M32, U8, U32, Memory, SimError, Wave, Workgroup
from .ops import compile_inst

HIDDEN_OK = {"hidden_block_count_x", "hidden_block_count_y", "hidden_block_count_z", "hidden_group_size_x", "hidden_group_size_y",
             "hidden_group_size_z", "hidden_remainder_x", "hidden_remainder_y", "hidden_remainder_z", "hidden_global_offset_x",
             "hidden_global_offset_y", "hidden_global_offset_z", "hidden_grid_dims", "hidden_dynamic_lds_size", "hidden_none",
             "hidden_printf_buffer", "hidden_hostcall_buffer", "hidden_multigrid_sync_arg", "hidden_heap_v1",
             "hidden_default_queue", "hidden_completion_action", "hidden_private_base", "hidden_shared_base", "hidden_queue_ptr"}


class Launch:
    def __init__(self, sim, kernel, grid, block, lds_dynamic, arg_bytes):
        self.sim = sim
        self.mem = sim.mem
        self.kernel = kernel
        self.grid, self.block = tuple(grid), tuple(block)
        self.lds_bytes = kernel.lds_static + lds_dynamic
        self.hazards = []
        self.strict = sim.strict
        self.ninst = 0
        self.stats = {} if sim.collect_stats else None     # mnemonic -> executed wave-instructions
        if kernel.scratch > 4096:
            raise SimError(f"{kernel.name}: {kernel.scratch} bytes of scratch per lane (only spill slots are modelled)")
        if kernel.preload & 0x7F:
            raise SimError("kernarg preload not modelled")
        if kernel.code_props & ~0x8 & 0x7F:
            raise SimError(f"{kernel.name}: user SGPRs {kernel.code_props:#x} beyond the kernarg pointer not modelled")
        # kernarg segment
        seg = np.zeros(max(kernel.kernarg_size, 8) + 64, U8)
        explicit = [a for a in kernel.args if not str(a[".value_kind"]).startswith("hidden_")]
        assert len(explicit) == len(arg_bytes), (kernel.name, len(explicit), len(arg_bytes))
        for a, raw in zip(explicit, arg_bytes):
            assert len(raw) == int(a[".size"])
            seg[int(a[".offset"]):int(a[".offset"]) + len(raw)] = np.frombuffer(raw, U8)
        for a in kernel.args:
            vk, off, size = str(a[".value_kind"]), int(a[".offset"]), int(a[".size"])
            if not vk.startswith("hidden_"):
                continue
            if vk not in HIDDEN_OK:
                raise SimError(f"hidden argument {vk} not modelled")
            val = None
            for i, ax in enumerate("xyz"):
                if vk == f"hidden_block_count_{ax}":
                    val = grid[i]
                elif vk == f"hidden_group_size_{ax}":
                    val = block[i]
                elif vk == f"hidden_remainder_{ax}":
                    val = 0
                elif vk == f"hidden_global_offset_{ax}":
                    val = 0
            if vk == "hidden_grid_dims":
                val = 3 if grid[2] > 1 else 2 if grid[1] > 1 else 1
            if vk == "hidden_dynamic_lds_size":
                val = lds_dynamic
            if val is not None:
                seg[off:off + size] = np.frombuffer(int(val).to_bytes(size, "little"), U8)
        self.kernarg = seg
        self.kernarg_addr = seg.ctypes.data
        assert self.kernarg_addr % 4 == 0
        self.mem.extra = [(self.kernarg_addr, seg)]

    def make_workgroup(self, gx, gy, gz):
        k = self.kernel
        nthreads = self.block[0] * self.block[1] * self.block[2]
        nwaves = (nthreads + 63) // 64
        wg = Workgroup(self, (gx, gy, gz), nwaves)
        tid = np.arange(nwaves * 64, dtype=np.int64)
        tx = tid % self.block[0]
        ty = (tid // self.block[0]) % self.block[1]
        tz = tid // (self.block[0] * self.block[1])
        packed = (tx | (ty << 10) | (tz << 20)).astype(U32)
        for i in range(nwaves):
            w = Wave(wg, i, k)
            w.s[0], w.s[1] = self.kernarg_addr & M32, self.kernarg_addr >> 32
            n = 2
            for bit, val in ((7, gx), (8, gy), (9, gz)):
                if (k.rsrc2 >> bit) & 1:
                    w.s[n] = val
                    n += 1
            w.v[0] = packed[64 * i:64 * i + 64]
            live = min(64, nthreads - 64 * i)
            w.set_exec((1 << live) - 1)
            w.pc = k.index[k.entry]
            wg.waves.append(w)
        return wg

    def run_workgroup(self, gx, gy, gz, order=0):
        wg = self.make_workgroup(gx, gy, gz)
        insts = self.kernel.insts
        waves = wg.waves
        at_barrier = [False] * len(waves)
        rounds = 0
        while True:
            alive = [i for i, w in enumerate(waves) if not w.done]
            if not alive:
                break
            seq = alive if order == 0 else alive[::-1] if order == 1 else list(np.random.RandomState(order + rounds).permutation(alive))
            for i in seq:
                if at_barrier[i]:
                    continue
                w = waves[i]
                r = self._run_wave(w, insts)
                if r == BARRIER:
                    at_barrier[i] = True
                else:
                    w.done = True
                    w.drain()
            alive = [i for i, w in enumerate(waves) if not w.done]
            if alive and all(at_barrier[i] for i in alive):
                for i in alive:
                    at_barrier[i] = False
            rounds += 1
        for w in waves:
            self.ninst += w.ninst
        if wg.lds_pending.any():
            raise SimError("LDS-DMA still pending at the end of the workgroup")
        return wg

    def _run_wave(self, w, insts):
        sim = self.sim
        limit = sim.max_inst
        stats = self.stats
        n = 0
        try:
            while True:
                ins = insts[w.pc]
                fn = ins.fn
                if fn is None:
                    fn = compile_inst(ins)
                if stats is not None:
                    stats[ins.mnem] = stats.get(ins.mnem, 0) + 1
                r = fn(w)
                n += 1
                if r is None:
                    w.pc += 1
                elif r >= 0:
                    w.pc = r
                else:
                    w.ninst += n
                    return r
                if n > limit:
                    raise SimError(f"wave exceeded {limit} instructions (runaway loop?) at {ins.addr:#x} `{ins.text}`")
        except SimError:
            raise
        except Exception as e:
            ins = insts[w.pc]
            raise SimError(f"{type(e).__name__}: {e} while executing {ins.addr:#x} `{ins.text}` (wave {w.wid}, workgroup {w.wg.id})") from e

    def run(self, order=0, select=None):
        gx, gy, gz = self.grid
        ids = [(x, y, z) for z in range(gz) for y in range(gy) for x in range(gx)]
        if select is not None:
            ids = [ids[i] for i in select]
        for (x, y, z) in ids:
            self.run_workgroup(x, y, z, order)


class Simulator:
    def __init__(self, lib=None, strict=False, check_bounds=True, max_inst=20_000_000, verbose=False, kernels=None, mem=None):
        """lib: a libbndm_hip.so (its code objects are parsed); or `kernels` (loader.load_code_object_file) + `mem` (a core.Memory over a
        private buffer) for stand-alone launches without the recording runtime"""
        self.kernels = kernels if kernels is not None else loader.load_library(lib)
        self.mem = mem if mem is not None else Memory()
        self.strict = strict or os.environ.get("GFX950SIM_STRICT") == "1"      # the first hazard ends the run
        self.check_bounds = check_bounds
        self.max_inst = max_inst
        self.verbose = verbose
        self.collect_stats = os.environ.get("GFX950SIM_STATS") == "1"
        self.stats = []                  # (kernel, grid, {mnemonic: count}, {memory counter: bytes}) per launch, single-process runs only
        self.order = int(os.environ.get("GFX950SIM_ORDER", "0"))
        self.nproc = int(os.environ.get("GFX950SIM_PROCS", "1"))
        self.log = []                    # (kernel, grid, instructions, seconds, hazards)
        self.hazards = []
        self._hook = None
        self.hip = None
        self.skip = set()                # kernel-name substrings not to execute
        self._tiny = set()
        self.subst = []                  # (substring, replacement, block.x, LDS bytes) kernel-variant substitutions
        for r in filter(None, os.environ.get("GFX950SIM_SUBST", "").split(";")):
            frm, rest = r.split("=>")
            to, bx, lds_ = rest.split(":")
            self.subst.append((frm, to, int(bx), int(lds_)))
        # sensitivity experiments: "substr:add" loosens every counted `s_waitcnt vmcnt(N > 0)` of the matching kernels by `add`
        # (tests/gfx950sim/mutate.py) -- a correct kernel must then FAIL here
        loosen = os.environ.get("GFX950SIM_LOOSEN")
        if loosen:
            sub, add = loosen.rsplit(":", 1)
            n = 0
            for name, k in self.kernels.items():
                if sub in name:
                    for ins in k.insts:
                        if ins.mnem == "s_waitcnt":
                            mods = []
                            for m in ins.mods:
                                if m.startswith("vmcnt(") and m != "vmcnt(0)":
                                    m = f"vmcnt({int(m[6:-1]) + int(add)})"
                                    n += 1
                                mods.append(m)
                            ins.mods, ins.fn = mods, None
            print(f"[sim] loosened {n} counted vmcnt waits in kernels matching {sub!r} by {add}", flush=True)
        self.reference = None            # differential mode: callable(name, grid, block, lds, args) running a model of the launch
        self.diffs = []                  # (kernel, launch index, report lines) of launches whose stores differ from the model's

    # --- the recording runtime's side ---------------------------------------------------------------------------------
    def install(self):
        self.hip = C.CDLL("libamdhip64.so.7")
        assert hasattr(self.hip, "hipmock_set_launch_hook"), "run under tests/hipmock's stand-in for libamdhip64.so.7"
        proto = C.CFUNCTYPE(None, C.c_char_p, C.POINTER(C.c_uint), C.c_size_t, C.POINTER(C.c_void_p))
        self._hook = proto(self._on_launch)
        self.hip.hipmock_set_launch_hook(self._hook)
        return self

    def uninstall(self):
        if self.hip is not None:
            self.hip.hipmock_set_launch_hook(None)

    def _refresh_allocs(self):
        if not self.check_bounds:
            self.mem.allocs = None
            return
        cap = 1 << 16
        ptrs, sizes = (C.c_uint64 * cap)(), (C.c_uint64 * cap)()
        n = self.hip.hipmock_allocs(ptrs, sizes, cap)
        self.mem.set_allocs([(int(ptrs[i]), int(sizes[i])) for i in range(n)])

    def _on_launch(self, name, dims, lds, args):
        try:
            name = name.decode()
            k = self.kernels.get(name)
            if k is None:
                raise SimError(f"launch of unknown kernel {name}")
            if any(s in name for s in self.skip):
                return
            dims = list(dims[0:6])
            for rule in self.subst:
                # run ANOTHER variant of the kernel on the recorded arguments (variants of one template share their argument
                # list; block size and LDS size are the variant's): how the 4-wave conv_t32 -- which the library only picks for
                # grids of >= 448 workgroups, i.e. batch >= 28 -- is executed on a batch the simulator can afford
                if rule[0] in name:
                    name = name.replace(rule[0], rule[1])
                    dims[3], lds = rule[2], rule[3]
                    k = self.kernels[name]
                    break
            explicit = [a for a in k.args if not str(a[".value_kind"]).startswith("hidden_")]
            raw = [C.string_at(args[i], int(a[".size"])) for i, a in enumerate(explicit)]
            self._refresh_allocs()
            if self.reference is not None:
                self._launch_differential(name, list(dims[0:3]), list(dims[3:6]), lds, raw)
            else:
                self.launch(name, dims[0:3], dims[3:6], lds, raw)
        except BaseException as e:          # an exception must not unwind through the C frames of the runtime
            import traceback
            traceback.print_exc()
            self.failed = e
            os._exit(97)

    def _launch_differential(self, name, grid, block, lds, raw):
        """the launch on the simulator, its stores undone, then the reference model of the same launch; the bytes the
        simulated kernel stored are compared with what the model left there.  Memory continues with the MODEL's result, so
        every launch is judged on clean inputs."""
        mem = self.mem
        only = os.environ.get("GFX950SIM_ONLY")
        self.nlaunch = getattr(self, "nlaunch", -1) + 1
        if only and str(self.nlaunch) not in only.split(",") and not any(t and not t.isdigit() and t in name for t in only.split(",")):
            self.reference(name, grid, block, lds, raw)
            self.log.append((name, tuple(grid), 0, 0.0, 0))
            return
        mem.undo = []
        nproc, self.nproc = self.nproc, 1
        # GFX950SIM_WGSAMPLE=n: only n workgroups of each launch (first, last, and spread in between) -- launches at the benchmark's
        # own sizes (batch 64: the large-index end of every address computation) for the price of a few workgroups each
        select = None
        ns = int(os.environ.get("GFX950SIM_WGSAMPLE", "0"))
        nwg = grid[0] * grid[1] * grid[2]
        if ns and nwg > ns:
            select = sorted({0, nwg - 1} | {int(round(i * (nwg - 1) / max(ns - 1, 1))) for i in range(ns)})[:max(ns, 2)]
        try:
            self.launch(name, grid, block, lds, raw, select=select)
        finally:
            self.nproc = nproc
        log, mem.undo = mem.undo, None
        if log:
            idx = np.unique(np.concatenate([i for i, _ in log]))
            sim_vals = mem.u8[idx].copy()
            for i, old in reversed(log):
                mem.u8[i] = old
        else:
            idx = np.zeros(0, np.int64)
            sim_vals = np.zeros(0, U8)
        self.reference(name, grid, block, lds, raw)
        ref_vals = mem.u8[idx]
        bad = sim_vals != ref_vals
        rep = [f"{loader_short(name)} launch #{len(self.log) - 1}: {int(bad.sum())} of {idx.size} stored bytes differ from the model"]
        if idx.size:
            starts, ends = mem.allocs
            a = idx + mem.base
            which = np.searchsorted(starts, a, side="right") - 1
            for wi in np.unique(which):
                sel = which == wi
                o = (a[sel] - starts[wi])
                sv, rv = sim_vals[sel], ref_vals[sel]
                line = f"   allocation {int(starts[wi]):#x} (+{int(o.min())}..{int(o.max())}, {int(sel.sum())} bytes): {int((sv != rv).sum())} differ"
                if (sv != rv).any():
                    # as 16-bit / 32-bit floats over the aligned elements whose bytes were all stored (the stored set need not be contiguous)
                    for dt, nb in ((np.float16, 2), (np.float32, 4)):
                        i0 = np.nonzero((o[:o.size - nb + 1] % nb == 0) & (o[nb - 1:] - o[:o.size - nb + 1] == nb - 1))[0]
                        if i0.size == 0:
                            continue
                        gi = i0[:, None] + np.arange(nb)
                        x = np.ascontiguousarray(sv[gi]).view(dt).ravel().astype(np.float64)
                        y = np.ascontiguousarray(rv[gi]).view(dt).ravel().astype(np.float64)
                        fin = np.isfinite(x) & np.isfinite(y)
                        rel = np.linalg.norm((x - y)[fin]) / max(np.linalg.norm(y[fin]), 1e-30)
                        line += f" | as {np.dtype(dt).name}: rel-L2 {rel:.3e}, max|d| {np.abs(x - y)[fin].max() if fin.any() else 0:.3e}, non-finite {int((~fin).sum())}"
                    k = int(np.nonzero(sv != rv)[0][0])
                    line += f" | first at +{int(o[k])}"
                rep.append(line)
        dump = os.environ.get("GFX950SIM_DUMP")
        if dump and bad.any():
            np.savez(os.path.join(dump, f"diff_{len(self.log) - 1}.npz"), addr=idx + mem.base, sim=sim_vals, ref=ref_vals)
        self.diffs.append((name, len(self.log) - 1, rep, int(bad.sum()), int(idx.size)))
        if self.verbose:
            print("\n".join(rep), flush=True)

    # --- direct use -----------------------------------------------------------------------------------------------------
    def launch(self, name, grid, block, lds_dynamic, arg_bytes, select=None):
        k = self.kernels[name]
        L = Launch(self, k, list(grid), list(block), lds_dynamic, arg_bytes)
        t0 = time.time()
        nwg = grid[0] * grid[1] * grid[2]
        if self.nproc > 1 and nwg >= 2 and select is None and L.kernel.name not in self._tiny:
            self._run_forked(L, nwg, min(self.nproc, nwg))
        else:
            L.run(self.order, select)
        dt = time.time() - t0
        if dt < 0.05 and self.nproc > 1:
            self._tiny.add(k.name)           # (a launch this short is not worth the forks next time)
        if L.stats is not None:
            self.stats.append((name, tuple(grid), L.stats, dict(self.mem.counters)))
            self.mem.counters.clear()
        self.log.append((name, tuple(grid), L.ninst, dt, len(L.hazards)))
        if os.environ.get("GFX950SIM_COVERAGE"):          # one line per executed launch: the kernel that RAN (after substitution)
            with open(os.environ["GFX950SIM_COVERAGE"], "a") as f:
                f.write(f"{k.name} {L.ninst} {len(L.hazards)}\n")
        self.hazards.extend(f"{name[:60]}: {h}" for h in L.hazards)
        if self.verbose:
            print(f"[sim] {loader_short(name):40s} grid={tuple(grid)} block={tuple(block)} {L.ninst:9d} wave-insts {dt:6.2f}s hazards={len(L.hazards)}",
                  flush=True)
        return L

    def _run_forked(self, L, nwg, nproc):
        """workgroups dealt to forked children; device memory is a MAP_SHARED region, so their stores land in place"""
        import pickle
        pids = []
        for p in range(nproc):
            r, wfd = os.pipe()
            pid = os.fork()
            if pid == 0:
                os.close(r)
                code = 0
                try:
                    self.mem.counters.clear()
                    L.run(self.order, range(p, nwg, nproc))
                    os.write(wfd, pickle.dumps((L.ninst, L.hazards, None, L.stats, dict(self.mem.counters))))
                except BaseException as e:
                    import traceback
                    os.write(wfd, pickle.dumps((L.ninst, L.hazards, traceback.format_exc()[-3000:], None, {})))
                    code = 1
                os._exit(code)
            os.close(wfd)
            pids.append((pid, r))
        err = None
        for pid, r in pids:
            buf = b""
            while True:
                chunk = os.read(r, 1 << 16)
                if not chunk:
                    break
                buf += chunk
            os.close(r)
            os.waitpid(pid, 0)
            if buf:
                n, hz, e, st, ctr = pickle.loads(buf)
                L.ninst += n
                L.hazards.extend(hz)
                err = err or e
                if st and L.stats is not None:
                    for k_, v_ in st.items():
                        L.stats[k_] = L.stats.get(k_, 0) + v_
                for k_, v_ in ctr.items():
                    self.mem.counters[k_] = self.mem.counters.get(k_, 0) + v_
            else:
                err = err or "worker died"
        if err:
            raise SimError("worker failed:\n" + err)


def loader_short(sym):
    import re
    m = re.match(r"^_ZN(?:4bndm)?(?:12_GLOBAL__N_1)?(\d+)", sym)
    if not m:
        return sym[:40]
    n = int(m.group(1))
    p = m.end()
    return sym[p:p + n] + sym[p + n:p + n + 28]
