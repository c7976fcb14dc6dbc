"""CPU tests of the host-side code around the C ABI (no GPU, no compute calls):
  * EvaluationDomain::new in both host mirrors (Python: halo2-pse_amd/__init__.py, C++: host/halo2hip.hpp) against the
    golden domain constants -- neither takes them from the oracle;
  * the C++ host code (tests/cpp/*.cpp with host/*.hpp and csrc/{field,fieldu,ec,ecu,glv}.h compiled for the host)
    under AddressSanitizer + UndefinedBehaviorSanitizer;
  * bench.py --gpus N started bare spawns its N ranks itself and hands their exit code through."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

DOMAINS = [(2, 3), (3, 4), (4, 5)]
FIELDS = ("omega", "omega_inv", "extended_omega", "extended_omega_inv", "g_coset", "g_coset_inv", "ifft_divisor",
          "extended_ifft_divisor")


def _limbs_from_hex(h):
    v = int(h, 16)
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


@pytest.mark.parametrize("jk", DOMAINS)
def test_python_evaluation_domain_new_matches_golden(h2, golden, jk):
    j, k = jk
    d = h2.EvaluationDomain.new(j, k)
    assert d.extended_k == int(np.ravel(golden[f"domain_{j}_{k}_extended_k"])[0])
    for f in FIELDS:
        assert np.array_equal(getattr(d, f), golden[f"domain_{j}_{k}_{f}"].reshape(4)), f
    assert np.array_equal(d.barycentric_weight, golden[f"domain_{j}_{k}_barycentric_weight"].reshape(4))
    assert np.array_equal(d.t_evaluations, golden[f"domain_{j}_{k}_t_evaluations"].reshape(-1, 4))


def test_python_domain_known_sizes(h2):
    # k: 5, extended_k: 7 is pinned in the reference's tests/plonk_api.rs:629-632 (j = 4 there); SURVEY.md 3.4: k = 17 -> 19
    assert h2.EvaluationDomain.new(4, 5).extended_k == 7
    assert h2.EvaluationDomain.new(4, 17).extended_k == 19
    d = h2.EvaluationDomain.new(2, 22)
    w = h2.fr_to_int(d.omega)
    assert pow(w, 1 << 21, h2.FR_MODULUS) == h2.FR_MODULUS - 1  # omega has order exactly 2^22
    assert h2.fr_to_int(d.omega) * h2.fr_to_int(d.omega_inv) % h2.FR_MODULUS == 1


@pytest.mark.parametrize("jk", DOMAINS)
def test_cpp_evaluation_domain_new_matches_golden(golden, jk):
    j, k = jk
    exe = os.path.join(ROOT, "tests", "cpp", "test_host_mirror")
    r = subprocess.run([exe, "--dump-domain", str(j), str(k)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    got, t_evals = {}, []
    for line in r.stdout.splitlines():
        name, val = line.split()
        if name == "t_evaluations":
            t_evals.append(_limbs_from_hex(val))
        elif name == "extended_k":
            got[name] = int(val)
        else:
            got[name] = _limbs_from_hex(val)
    assert got["extended_k"] == int(np.ravel(golden[f"domain_{j}_{k}_extended_k"])[0])
    for f in FIELDS + ("barycentric_weight",):
        assert np.array_equal(got[f], golden[f"domain_{j}_{k}_{f}"].reshape(4)), f
    assert np.array_equal(np.stack(t_evals), golden[f"domain_{j}_{k}_t_evaluations"].reshape(-1, 4))


def test_host_code_under_asan_ubsan(tmp_path):
    """The host-side arithmetic and the C++ mirror, built with -fsanitize=address,undefined (CPU only; GPU sanitizers are
    not available on the pool).  test_fieldu fuzzes fieldu/ecu/glv against field/ec; test_host_mirror's CPU-only mode
    walks EvaluationDomain::new and the evaluate_h graph builders."""
    inc = ["-I" + os.path.join(ROOT, "halo2-pse_amd", "csrc"), "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "halo2-pse_amd")]
    san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1")
    exe = str(tmp_path / "test_fieldu_san")
    src = os.path.join(ROOT, "tests", "cpp", "test_fieldu.cpp")
    subprocess.check_call(["g++", "-std=c++17", "-DH2_FU_CHECK", "-Wno-unknown-pragmas"] + san + inc + [src, "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "fieldu tests ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    exe2 = str(tmp_path / "test_host_mirror_san")
    src2 = os.path.join(ROOT, "tests", "cpp", "test_host_mirror.cpp")
    libdir = os.path.join(ROOT, "halo2-pse_amd")
    subprocess.check_call(["g++", "-std=c++17", "-Wno-unknown-pragmas"] + san + inc + [src2, "-o", exe2, "-L" + libdir, "-lhalo2hip",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    for args in (["--dump-domain", "4", "5"], ["--dump-domain", "2", "10"], ["--dump-graphs", str(tmp_path / "graphs.bin")]):
        r = subprocess.run([exe2] + args, capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, (args, r.stdout[-2000:], r.stderr[-4000:])
        assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]


def test_engine_host_logic_under_tsan(tmp_path):
    """the multi-device host logic of csrc/api.hip -- per-context locks under the shared engine lock, the
    single-slot hand-off to the per-device Worker threads, the copier threads of the host-pointer batched transforms, init /
    shutdown racing with callers, the cached no-GPU state -- never ran on two distinct GPUs.  Here api.hip is compiled as plain
    C++ against a stub HIP runtime with two fake devices (tests/cpp/hipstub) and driven from a dozen threads under
    ThreadSanitizer (tests/cpp/test_engine_tsan.cpp); TSan turns any report into a non-zero exit."""
    csrc = os.path.join(ROOT, "halo2-pse_amd", "csrc")
    cpp = os.path.join(ROOT, "tests", "cpp")
    exe = str(tmp_path / "test_engine_tsan")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-Wno-unknown-pragmas", "-I" + os.path.join(cpp, "hipstub"), "-I" + csrc,
                           "-x", "c++", os.path.join(csrc, "api.hip"), os.path.join(cpp, "engine_stubs.cpp"), os.path.join(cpp, "test_engine_tsan.cpp"),
                           "-o", exe, "-lpthread", "-ldl"])
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66 second_deadlock_stack=1")
    env.pop("H2_STUB_DEVICES", None)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "engine host logic under tsan: ok" in r.stdout, r.stdout[-3000:] + r.stderr[-6000:]
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-6000:]


def test_engine_host_logic_under_asan(tmp_path):
    """The same program (api.hip on the stub HIP runtime, a dozen threads through init / shutdown, the copier queues, the pinned-column
    cache and the per-device workers) under AddressSanitizer + UndefinedBehaviorSanitizer with leak detection on: the host logic neither
    touches freed memory nor leaks across h2hip_shutdown."""
    csrc = os.path.join(ROOT, "halo2-pse_amd", "csrc")
    cpp = os.path.join(ROOT, "tests", "cpp")
    exe = str(tmp_path / "test_engine_asan")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
                           "-Wno-unknown-pragmas", "-I" + os.path.join(cpp, "hipstub"), "-I" + csrc,
                           "-x", "c++", os.path.join(csrc, "api.hip"), os.path.join(cpp, "engine_stubs.cpp"), os.path.join(cpp, "test_engine_tsan.cpp"),
                           "-o", exe, "-lpthread", "-ldl"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("H2_STUB_DEVICES", None)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "engine host logic under tsan: ok" in r.stdout, r.stdout[-3000:] + r.stderr[-6000:]
    assert "AddressSanitizer" not in r.stderr and "LeakSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-6000:]


def test_bench_spawns_its_ranks_when_started_bare():
    """`python bench.py --gpus 2` with no WORLD_SIZE must start two ranks itself (the driver's launch shape) instead of
    exiting with a usage error.  There is no GPU here, so both ranks stop with the no-GPU message and the parent hands the
    non-zero exit code through; what is checked is that they were started."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu-marked bench test")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    out = r.stdout + r.stderr
    assert r.returncode != 0
    assert "launch with torch.distributed.run" not in out
    # both ranks were started: each either printed the no-GPU message or was torn down by the launcher once its peer had
    assert out.count("bench.py needs a GPU") >= 1 and ("local_rank: 1" in out or out.count("bench.py needs a GPU") >= 2), out[-3000:]


def test_domain_new_matches_the_reference_pinned_pasta_omega(h2):
    """The one constant of this path's logic that the reference's own tests hold as bytes: the verification key pinned in
    tests/plonk_api.rs:624-632 (curve EqAffine, scalar field pasta Fp) records `k: 5, extended_k: 7, omega: 0x0cc3...78cc`.
    EvaluationDomain.new is generic over the field as the reference's is (poly/domain.rs:39-142); run over pasta Fp
    (generator 5, S = 32) with j = 5 it must reproduce exactly that.  (BN254 itself stays "parity unpinned": the reference
    holds no BN254 bytes.)"""
    d = h2.EvaluationDomain.new(5, 5, field=h2.PASTA_FP)
    assert d.k == 5 and d.extended_k == 7
    assert d.ints["omega"] == 0x0cc3380dc616f2e1daf29ad1560833ed3baea3393eceb7bc8fa36376929b78cc
    p = h2.PASTA_FP.modulus
    assert p == 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001  # `scalar_modulus` of the same pinned key (:627)
    assert pow(d.ints["omega"], 32, p) == 1 and pow(d.ints["omega"], 16, p) == p - 1
    assert d.ints["omega"] * d.ints["omega_inv"] % p == 1 and d.ints["ifft_divisor"] * 32 % p == 1
    assert pow(d.ints["extended_omega"], 4, p) == d.ints["omega"]
    # Montgomery limbs of the generic path agree with the BN254-only helper on BN254
    assert np.array_equal(h2.BN254_FR.from_int(12345), h2.fr_from_int(12345))
    with pytest.raises(h2.H2HipError):  # the device entry points serve bn256::Fr only
        d.lagrange_to_coeff(np.zeros((32, 4), dtype=np.uint64))


def test_stream_chunk_ladder_host_logic(h2):
    """the chunk ladder of a streamed host-slice MSM (msm.hip stream_ladder; no GPU involved): the pieces cover n exactly, all
    but the last are multiples of the 4096-scalar count tile, they grow by ~1/ratio for ratio < 1 (every upload ends as the
    previous chunk's work does) and shrink for ratio > 1 (upload-bound: the last chunk's work is the exposed part), small
    inputs collapse to fewer pieces, and below 2^20 pairs the default is two"""
    import ctypes
    L = h2.lib()
    L.h2hip_debug_msm_stream_ladder.restype = ctypes.c_size_t

    def ladder(n, chunks=0, permille=0, with_bases=0):
        out = (ctypes.c_size_t * 32)()
        k = L.h2hip_debug_msm_stream_ladder(ctypes.c_size_t(n), ctypes.c_uint32(chunks), ctypes.c_uint32(permille), ctypes.c_int(with_bases), out, ctypes.c_size_t(32))
        return list(out[:k])

    for n in (1 << 20, (1 << 20) + 12345, 1 << 24, (1 << 26) - 1, 600001):
        for chunks, pm in ((0, 0), (2, 600), (3, 600), (4, 450), (7, 900), (5, 1500)):
            sz = ladder(n, chunks, pm)
            assert sum(sz) == n and all(m > 0 for m in sz), (n, chunks, pm, sz)
            assert all(m % 4096 == 0 for m in sz[:-1])
            assert len(sz) <= (chunks or 3)
            if pm and pm < 1000 and len(sz) > 2:
                assert all(sz[i + 1] > sz[i] for i in range(len(sz) - 2)), sz  # growing (the last takes the remainder)
            if pm > 1000 and len(sz) > 2:
                assert all(sz[i + 1] < sz[i] for i in range(len(sz) - 2)), sz
    assert len(ladder(1 << 20)) == 3 and len(ladder((1 << 20) - 1)) == 2          # defaults: three chunks from 2^20, two below
    sz = ladder(1 << 20)
    assert abs(sz[0] / (1 << 20) - 0.184) < 0.01 and abs(sz[2] / (1 << 20) - 0.51) < 0.01   # 1 : 1/0.6 : 1/0.36
    assert len(ladder(1 << 20, with_bases=1)) == 6                                   # unpinned: twice the chunks at twice the ratio
    assert ladder(5000, 4, 600) == [5000] and len(ladder(20000, 4, 600)) == 2        # too small to cut that often
    assert len(ladder(40123, 7, 900)) == 4 and len(ladder(600001, 7, 900)) == 7      # an explicit chunk count is taken as given
    assert ladder(0, 3, 600) == [0]
