"""CPU-only tests: host logic, config files, C ABI surface, multi-rank result gather (gloo)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_library_exports_every_declared_symbol():
    from visual_odometry_amd import _abi

    hdr = open(os.path.join(ROOT, "include", "vo_mi355x.h")).read()
    names = sorted(set(re.findall(r"\b(vo_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 25
    lib = _abi.load_library()
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/vo_mi355x.h but not exported"
    assert lib.vo_abi_version() == 1


def test_struct_layouts_match_header():
    from visual_odometry_amd import _abi

    assert C.sizeof(_abi.VoKeypoint) == 28 and C.sizeof(_abi.VoDMatch) == 16
    assert C.sizeof(_abi.VoPairOut) == 10 * 4 + 21 * 8
    assert C.sizeof(_abi.VoPairIn) == 4 * 8 + 2 * 4 + 9 * 8  # vo_pair_in of vo_pair_batch
    assert _abi.VoPairIn.rows.offset == 32 and _abi.VoPairIn.K.offset == 40
    lib = _abi.load_library()
    p = _abi.VoParams()
    lib.vo_params_default(C.byref(p))
    d = _abi.default_params()
    for f, _ in _abi.VoParams._fields_:
        assert getattr(p, f) == getattr(d, f), f
    # reference constants (camera_utility.hpp:33-63)
    assert (p.hessian_threshold, p.n_octaves, p.n_octave_layers, p.extended, p.upright) == (50.0, 4, 3, 0, 0)
    assert abs(p.ratio_thresh - 0.7) < 1e-7 and (p.roi_x_lo, p.roi_y_lo, p.roi_x_hi, p.roi_y_hi) == (28, 20, 614, 482)
    assert (p.ransac_prob, p.ransac_threshold, p.ransac_max_iters, p.ransac_seed) == (0.99, 2.0, 1000, 2**64 - 1)
    assert abs(p.inlier_threshold - 0.3) < 1e-7 and abs(p.vpf_threshold - 0.85) < 1e-7
    assert p.distance_threshold == 50.0 and p.reprojection_tolerance == 0.5


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_no_cpu_fallback_without_gpu():
    """The product has no CPU path: without a gfx950 device the context cannot be created."""
    from visual_odometry_amd.frontend import Context, VoError

    with pytest.raises(VoError) as e:
        Context()
    assert e.value.status == -2


def test_sample_stream_host_entry_matches_oracle():
    import oracle_lib as O
    from visual_odometry_amd import _abi

    lib = _abi.load_library()
    for seed, count, nt in [(2**64 - 1, 8192, 4096), (2**64 - 1, 37, 300), (0, 100, 50), (7, 6, 100)]:
        out = np.zeros((nt, 5), np.int32)
        assert lib.vo_ransac_subsets(C.c_uint64(seed), count, nt, out.ctypes.data_as(C.POINTER(C.c_int32))) == 0
        assert np.array_equal(out, O.ransac_subsets(seed, count, nt))
        assert out.min() >= 0 and out.max() < count
        assert all(len(set(r)) == 5 for r in out.tolist())


def test_camera_config_files():
    from visual_odometry_amd.config import CameraConfig, load_camera_config

    cfg_dir = os.path.join(ROOT, "config")
    par = load_camera_config(os.path.join(cfg_dir, "camera_intrinsic.yaml"))
    assert par.fx == par.fy == 407.0646129842357 and (par.cx, par.cy) == (384.5, 246.5)
    assert par.skew == 0 and par.desired_width is None  # file has neither key
    assert np.array_equal(par.K_for(640, 480), par.K())
    rap = load_camera_config(os.path.join(cfg_dir, "rapallo_intrinsic.yaml"))
    assert (rap.desired_width, rap.desired_height) == (640, 505)
    assert rap.k1 == 0.0626067211793766 and rap.p2 == -0.00554890468657758
    K = rap.K_for(2560, 2056)
    assert np.isclose(K[0, 0], rap.fx * 640 / 2560) and np.isclose(K[1, 2], rap.cy * 505 / 2056)
    assert isinstance(par, CameraConfig)


def test_pair_sharding_is_a_partition():
    from visual_odometry_amd.shard import pairs_of_rank

    for n, w in [(16, 1), (16, 8), (13, 4), (3, 8)]:
        allp = sorted(sum((pairs_of_rank(n, r, w) for r in range(w)), []))
        assert allp == list(range(n))
    with pytest.raises(ValueError):
        pairs_of_rank(4, 2, 2)


_GLOO_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch.distributed as dist
from visual_odometry_amd.shard import pairs_of_rank, gather_records, RECORD_FIELDS, RECORD_DOUBLES
rank, world, n_pairs = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), 7
dist.init_process_group("gloo", rank=rank, world_size=world)
mine = pairs_of_rank(n_pairs, rank, world)
local = np.array([[p * 10 + c for c in range(len(RECORD_FIELDS) + RECORD_DOUBLES)] for p in mine], np.float64).reshape(len(mine), -1)
table = gather_records(local, n_pairs, rank, world)
expect = np.array([[p * 10 + c for c in range(len(RECORD_FIELDS) + RECORD_DOUBLES)] for p in range(n_pairs)], np.float64)
assert np.array_equal(table, expect), (rank, table[:, 0])
dist.barrier(); dist.destroy_process_group()
print("ok", rank)
"""


def test_result_gather_two_ranks_gloo(tmp_path):
    """N>1 path of bench.py / offline replay: pairs sharded over ranks, one all_gather of the result records."""
    script = tmp_path / "w.py"
    script.write_text(_GLOO_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs


def test_cv_adaptor_and_ros_switch_headers_compile(tmp_path):
    """N3: include/vo_cv_adaptor.hpp + integration/ros/vo_mi355x_switch_*.hpp, used the way the node would use them
    (tests/cv_adaptor_check.cpp), compile and link against the library with the mock cv:: types of tests/mock_cv."""
    import subprocess

    exe = str(tmp_path / "cv_adaptor_check")
    lib_dir = os.path.join(ROOT, "visual_odometry_amd")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "tests", "mock_cv"),
                           "-I" + os.path.join(ROOT, "integration", "ros"), os.path.join(ROOT, "tests", "cv_adaptor_check.cpp"), "-o", exe,
                           "-L" + lib_dir, "-lvo_mi355x", "-Wl,-rpath," + lib_dir])
    assert subprocess.call([exe]) == 0  # without arguments: no GPU call, just proves the link


def test_matcher_device_code_has_no_mfma_read_hazard():
    """The matcher reads MFMA accumulators from inline assembly, which the compiler's hazard recogniser does not cover:
    tools/check_mfma_hazard.py compiles vo_match.hip to gfx950 assembly and checks the wait states itself."""
    import subprocess

    assert subprocess.call([sys.executable, os.path.join(ROOT, "tools", "check_mfma_hazard.py")]) == 0


def test_detection_device_code_has_no_sgpr_hazard_at_inline_assembly():
    """An SGPR written by a VALU instruction needs two wait states before a VALU instruction reads it on gfx950, and the compiler
    does not look inside inline assembly: tools/check_sgpr_hazard.py scans the compiled vo_detect.hip for it (and the scanner
    itself is checked on a hand-written violation first)."""
    import importlib.util, subprocess

    spec = importlib.util.spec_from_file_location("check_sgpr_hazard", os.path.join(ROOT, "tools", "check_sgpr_hazard.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    asm = "\tv_readfirstlane_b32 s4, v1\n\tv_readfirstlane_b32 s5, v2\n{gap}\t;;#ASMSTART\n\tv_lshl_add_u64 v[0:1], v[0:1], 0, s[4:5]\n\t;;#ASMEND\n"
    bad, n = mod.scan(asm.format(gap=""))
    assert n == 1 and len(bad) == 2  # s5 written 0 and s4 written 1 wait state earlier
    bad, n = mod.scan(asm.format(gap="\ts_nop 1\n"))
    assert n == 1 and bad == []
    bad, n = mod.scan(asm.format(gap="\tv_add_f32_e32 v3, v3, v3\n\tv_add_f32_e32 v3, v3, v3\n"))
    assert bad == []
    assert subprocess.call([sys.executable, os.path.join(ROOT, "tools", "check_sgpr_hazard.py")]) == 0


def test_bench_starts_its_own_ranks_for_gpus_n(monkeypatch):
    """`python bench.py --gpus N` outside torchrun: N ranks are started as CHILD processes through torch.distributed.run on
    127.0.0.1 (this process has not imported torch or touched HIP at that point), with the caller's arguments passed on."""
    import importlib
    import subprocess
    import sys

    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("RANK", raising=False)
    torch_loaded_before = "torch" in sys.modules
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7  # the children's exit code is this process's
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert ("torch" in sys.modules) == torch_loaded_before  # launching did not pull torch into the parent


def test_pair_batch_argument_checks_need_no_gpu():
    """vo_pair_batch rejects bad arguments before any thread or HIP call; vo_device_count is 0 without a device."""
    from visual_odometry_amd import _abi

    lib = _abi.load_library()
    outs = (_abi.VoPairOut * 1)()
    ins = (_abi.VoPairIn * 1)()
    assert lib.vo_pair_batch(None, 1, ins, 1, outs) == -1
    none = (C.c_void_p * 1)(None)
    assert lib.vo_pair_batch(none, 1, ins, 1, outs) == -1
    if not os.path.exists("/dev/kfd"):
        assert lib.vo_device_count() == 0
