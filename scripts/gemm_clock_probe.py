"""Does the GEMM main loop lose its MFMA / operand-fetch overlap to DVFS?  Runs one 8192^3 GEMM variant (production library and the
diagnostic builds in scripts/ubench/bin, see scripts/ab_gemm.py) back-to-back for ~1.5 s each while a thread samples the shader clock and
the socket power (amd-smi / rocm-smi / sysfs, whichever answers)."""
import ctypes
import glob
import os
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import audiolm_pytorch_amd  # noqa: E402,F401
from audiolm_pytorch_amd import _lib  # noqa: E402


def read_sysfs():
    out = {}
    for f in glob.glob('/sys/class/drm/card*/device/pp_dpm_sclk'):
        try:
            cur = [l for l in open(f).read().splitlines() if l.strip().endswith('*')]
            if cur:
                out['sclk'] = cur[0].split(':')[1].strip().rstrip('*').strip()
        except Exception:
            pass
    for f in glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*/power1_average') + glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*/power1_input'):
        try:
            out['power_W'] = int(open(f).read()) / 1e6
        except Exception:
            pass
    for f in glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input'):
        try:
            out['freq1_MHz'] = int(open(f).read()) / 1e6
        except Exception:
            pass
    return out


def main():
    dev = torch.device('cuda')
    libs = [('prod', _lib.load())]
    for p in sorted(glob.glob(os.path.join(ROOT, 'scripts/ubench/bin/libgemm_v*.so'))):
        lib = ctypes.CDLL(p)
        lib.alm_gemm_bf16_nt_tile.argtypes = _lib.SIGNATURES['alm_gemm_bf16_nt_tile']
        lib.alm_gemm_bf16_nt_tile.restype = ctypes.c_int
        libs.append((os.path.basename(p)[3:-3], lib))
    M = N = K = 8192
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    B = torch.randn(N, K, device=dev).to(torch.bfloat16)
    C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    print('sysfs probe:', read_sysfs(), flush=True)
    for tool in (['amd-smi', 'metric', '-c', '-p'], ['rocm-smi', '-c', '-P']):
        try:
            r = subprocess.run(tool, capture_output=True, text=True, timeout=20)
            print(' '.join(tool), '->', r.returncode, r.stdout[-600:].replace('\n', ' | '), flush=True)
        except Exception as e:
            print(' '.join(tool), 'unavailable:', e, flush=True)
    for name, lib in libs:
        samples, stop = [], threading.Event()

        def poll():
            while not stop.is_set():
                samples.append(read_sysfs())
                time.sleep(0.05)
        th = threading.Thread(target=poll)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        th.start()
        t0 = time.time()
        n = 0
        e0.record()
        while time.time() - t0 < 1.5:
            for _ in range(50):
                lib.alm_gemm_bf16_nt_tile(A.data_ptr(), B.data_ptr(), C.data_ptr(), None, M, N, K, K, K, N, 1.0, 0, 0, 2, st)
            n += 50
            torch.cuda.synchronize()
        e1.record()
        torch.cuda.synchronize()
        stop.set()
        th.join()
        us = e0.elapsed_time(e1) / n * 1e3
        keys = sorted({k for s in samples for k in s})
        summ = {k: sorted({str(s.get(k)) for s in samples[len(samples) // 3:]}) for k in keys}
        print(f'{name:16s} {us:8.1f} us/launch  {summ}', flush=True)


if __name__ == '__main__':
    main()
