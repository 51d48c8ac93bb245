"""The drop-in boundary from plain C: examples/c_client.c (C99, only include/fastmatch_hip.h and libfastmatch_hip.so)
compiles without warnings and, on a machine without a device, reports that instead of computing anything on the CPU.
On the GPU (test_c_client_gpu below) its output is the oracle's."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_client(tmp_path):
    exe = str(tmp_path / "c_client")
    lib_dir = os.path.join(ROOT, "fast-match_amd")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "c_client.c"), "-L", lib_dir, "-lfastmatch_hip",
                           "-Wl,-rpath," + lib_dir, "-lm", "-o", exe])
    return exe


def client_data(nq, nt, seed):
    """The rows examples/c_client.c generates (its LCG, its planted near-copies)."""
    state = seed & 0xFFFFFFFF

    def fill(n):
        nonlocal state
        out = np.empty(n * 128, dtype=np.uint8)
        for i in range(n * 128):
            state = (state * 1664525 + 1013904223) & 0xFFFFFFFF
            r = state >> 8
            out[i] = 40 + (r >> 4) % 120 if (r & 7) == 0 else (r >> 4) % 48
        return out.reshape(n, 128)

    Q, T = fill(nq), fill(nt)
    flip = (np.arange(128) % 17 == 0).astype(np.uint8)
    for i in range(0, min(nq, nt), 3):
        T[i] = Q[i] ^ flip
    return Q, T


def test_c_client_builds_and_refuses_to_run_without_a_device(tmp_path):
    import torch
    exe = build_client(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    p = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert p.returncode == 2 and "no CPU fallback" in p.stderr and p.stdout == ""


@pytest.mark.gpu
@pytest.mark.parametrize("nq,nt,seed", [(300, 120, 12345), (1, 1, 7), (33, 700, 99), (1500, 1400, 2024)])
def test_c_client_gpu(tmp_path, nq, nt, seed):
    import oracle
    exe = build_client(tmp_path)
    p = subprocess.run([exe, str(nq), str(nt), str(seed)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    lines = p.stdout.strip().splitlines()
    Q, T = client_data(nq, nt, seed)
    tidx, dist = oracle.bf_xcheck1(Q, T)
    sd = oracle.self_dist(Q)
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = dist.astype(np.float64) / sd
    acc = [q for q in range(nq) if tidx[q] >= 0 and ratio[q] < 0.9]
    assert lines[0] == "accepted %d of %d" % (len(acc), nq)
    assert len(lines) == 1 + len(acc)
    for line, q in zip(lines[1:], acc):
        f = line.split()
        assert (int(f[0]), int(f[1])) == (q, int(tidx[q]))
        assert np.float32(f[2]) == dist[q] and float(f[3]) == ratio[q]
