"""A/B of library variants on ONE box with power and clock beside the time (round 5; successor of tc_ab.py).

For every spec, in the order given, a fresh process times LAUNCHES calls of the default front end on 16 384 windows (HIP events of
the library: ms per 4096 windows) while a helper thread samples board power and shader clock from the amdgpu hwmon files every
20 ms (rocm-smi as the fallback), and hashes scores + intermediates of 600 windows so that a schedule change shows that the bits
did not move.  Output per spec: ms per 4096 windows, mean W, mean MHz, mJ per window (= W x s / windows), bits.

    tc_ab2.py [--rounds R] [--launches L] spec ...      spec = NAME | LIB.so | LIB.so,ENV=VAL[,ENV=VAL]   ("default" = in-tree library)
"""
import glob
import hashlib
import os
import subprocess
import sys
import threading
import time


def hwmon_files():
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        p = next((os.path.join(d, f) for f in ("power1_average", "power1_input") if os.path.exists(os.path.join(d, f))), None)
        if p:
            f = os.path.join(d, "freq1_input")
            return p, (f if os.path.exists(f) else None)
    return None, None


class Sampler:
    def __init__(self):
        self.watts, self.mhz, self._stop = [], [], threading.Event()
        self.pfile, self.ffile = hwmon_files()
        self._t = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        import re
        while not self._stop.is_set():
            try:
                if self.pfile:
                    self.watts.append(int(open(self.pfile).read()) / 1e6)
                    if self.ffile:
                        self.mhz.append(int(open(self.ffile).read()) / 1e6)
                    self._stop.wait(0.02)
                else:
                    out = subprocess.run(["rocm-smi", "-d", "0", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
                    m = re.search(r"Power \(W\):\s*([0-9.]+)", out)
                    if m:
                        self.watts.append(float(m.group(1)))
                    m = re.search(r"sclk clock level.*\((\d+)Mhz\)", out)
                    if m:
                        self.mhz.append(float(m.group(1)))
            except Exception:  # noqa: BLE001
                self._stop.wait(0.05)

    def __enter__(self):
        self._t.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        self._t.join(timeout=15)


if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import numpy as np
    sys.path.insert(0, ".")
    from genomad_amd import synthetic, _lib
    from genomad_amd.engine import NNEngine
    launches = int(sys.argv[2])
    eng = NNEngine(0, synthetic.synth_weights(), chunk=16384)
    N = 16384
    b, s = eng.alloc(N * 6000), eng.alloc(N * 12)
    eng.synth_windows_dev(0, N, b.ptr)
    for _ in range(3):
        eng.classify_dev(b.ptr, N, s.ptr, "f16x3tc")
    eng.sync()
    eng.profile_enable(True)
    eng.profile_reset()
    with Sampler() as sm:
        t0 = time.perf_counter()
        for _ in range(launches):
            eng.classify_dev(b.ptr, N, s.ptr, "f16x3tc")
        eng.sync()
        wall = time.perf_counter() - t0
    ms, l = eng.profile_get(_lib.K_FUSED)
    eng.profile_enable(False)
    idle = []
    time.sleep(0.5)
    with Sampler() as si:
        time.sleep(0.3)
    h = hashlib.md5(s.download((N, 3), np.float32).tobytes())
    wins = synthetic.synth_windows(5, 600)
    wins[7, 3000:] = 4
    wins[11, :] = 4
    sc, taps = eng.debug_forward(wins, "f16x3tc", taps=("m_a", "m_b", "yp_a", "yp_b"))
    exact = eng.classify(wins, "f32")
    h.update(sc.tobytes())
    for k in ("m_a", "m_b", "yp_a", "yp_b"):
        h.update(taps[k].tobytes())
    # per-phase cycle counters of the instrumented build of THIS library (every variant carries fused_front_tc_kernel<true>): the
    # critical path C of a 96-row step on both roles - what the time model T ~ C^a W^(1-a) needs beside the launch time
    import ctypes as C_
    _lib.check(eng.lib.gnn_phase_cycles(eng.ctx, 1, None))
    eng.classify_dev(b.ptr, 4096, s.ptr, "f16x3tc")
    eng.sync()
    out = (C_.c_uint64 * 16)()
    _lib.check(eng.lib.gnn_phase_cycles(eng.ctx, 0, out))
    per = [v / (4096 * 63) for v in out]
    cyc = f"cycles/step matrix {sum(per[:8]):6.0f} [{' '.join(f'{v:.0f}' for v in per[:8])}] helper {sum(per[8:]):6.0f} [{' '.join(f'{v:.0f}' for v in per[8:])}]"
    w = sm.watts[len(sm.watts) // 5:] or [float("nan")]          # drop the ramp
    f = sm.mhz[len(sm.mhz) // 5:] or [float("nan")]
    wm, fm = sum(w) / len(w), sum(f) / len(f)
    wi = sum(si.watts) / len(si.watts) if si.watts else float("nan")
    print(f"{ms / l / (N // 4096):8.3f} ms/4096   {wm:7.1f} W ({len(w)} samples, idle after {wi:6.1f})  {fm:6.0f} MHz   "
          f"{wm * (ms / l * 1e-3) / N * 1e3:6.3f} mJ/window   front/wall {ms / 1e3 / wall:5.3f}   dscore {np.abs(sc - exact).max():.2e}   bits {h.hexdigest()[:12]}   {cyc}")
    sys.exit(0)

args = sys.argv[1:]
rounds, launches = 1, 12
while args and args[0].startswith("--"):
    if args[0] == "--rounds":
        rounds = int(args[1])
    elif args[0] == "--launches":
        launches = int(args[1])
    args = args[2:]
for r in range(rounds):
    for spec in args:
        parts = spec.split(",")
        env = {k: v for k, v in os.environ.items() if k != "GENOMAD_AMD_LIB"}
        if parts[0] != "default":
            env["GENOMAD_AMD_LIB"] = parts[0]
        for kv in parts[1:]:
            k, v = kv.split("=", 1)
            env[k] = v
        try:
            out = subprocess.run([sys.executable, __file__, "--child", str(launches)], env=env, capture_output=True, text=True, timeout=300)
            text = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else "FAILED: " + out.stderr[-400:].replace("\n", " | ")
        except subprocess.TimeoutExpired:
            text = "TIMEOUT"
        print(f"{os.path.basename(spec):58s} {text}", flush=True)
