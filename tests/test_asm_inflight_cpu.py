"""CPU: static check of every compiled kernel for reads of registers whose hand-issued (`asm volatile`) LDS load is still in
flight (tools/check_asm_inflight.py).  hipcc cannot see that an asm ds_read delivers later, so a register copy it inserts for
an in/out asm operand, or a phi move, can land between the read and its `s_waitcnt` -- round 5 found one wave in ~40 launches of
the 64-wide 1x1 weight-gradient kernel multiplying a stale fragment that way.  hipcc cross-compiles without a GPU."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cvpr2023-unidistill_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# the translation units that issue LDS reads from inline asm
UNITS = ["conv2d_f32_wgrad", "conv2d_f32_1x1p", "conv2d_f32_wino4", "conv2d_f32_wino4_wgrad", "conv2d_f32_wino", "conv2d_f32_wino_wgrad",
         "conv2d", "spconv_conv"]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_no_kernel_reads_a_register_with_an_asm_load_in_flight(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_asm_inflight

    def compile_unit(name):
        out = str(tmp_path / (name + ".s"))
        extra = ["-fno-slp-vectorize"] if name == "conv2d_f32_wino4" else []          # as in the Makefile
        cmd = [HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", *extra,
               "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-S", "--cuda-device-only",
               os.path.join(CSRC, name + ".hip"), "-o", out]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        return out

    units = [u for u in UNITS if "asm volatile(\"ds_read" in open(os.path.join(CSRC, u + ".hip")).read()
             or "ds_read_b" in open(os.path.join(CSRC, u + ".hip")).read()]
    assert units
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        files = list(ex.map(compile_unit, units))
    bad = sum(check_asm_inflight.check(f) for f in files)
    assert bad == 0, f"{bad} instruction(s) read a register whose asm-issued LDS load is still in flight"


def test_checker_sees_the_hazard(tmp_path):
    """The pattern the checker exists for: a copy of an asm-loaded register before the wait that covers the load."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_asm_inflight
    bad_s = tmp_path / "bad.s"
    bad_s.write_text("_Zk:\n\t;;#ASMSTART\n\tds_read_b32 v17, v0 offset:0x1c00\n\t;;#ASMEND\n\tv_mov_b32_e32 v8, v17\n"
                     "\ts_waitcnt lgkmcnt(0)\n\ts_endpgm\n")
    ok_s = tmp_path / "ok.s"
    ok_s.write_text("_Zk:\n\t;;#ASMSTART\n\tds_read_b32 v17, v0 offset:0x1c00\n\t;;#ASMEND\n\t;;#ASMSTART\n\tds_read_b32 v18, v0\n"
                    "\t;;#ASMEND\n\ts_waitcnt lgkmcnt(1)\n\tv_mov_b32_e32 v8, v17\n\ts_waitcnt lgkmcnt(0)\n\tv_mov_b32_e32 v9, v18\n\ts_endpgm\n")
    assert check_asm_inflight.check(str(bad_s)) == 1
    assert check_asm_inflight.check(str(ok_s)) == 0


def test_checker_follows_branches_not_the_listing_order(tmp_path):
    """hipcc rotates loops: a body can be entered in the middle of the listing (round 5: the steady-state loop of the persistent
    1x1 kernel, whose second half precedes its entry block).  The checker walks the control-flow graph."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_asm_inflight
    rd = "\t;;#ASMSTART\n\tds_read_b32 v17, v0\n\t;;#ASMEND\n"
    hazard = tmp_path / "hazard.s"      # the read of v17 is reached through a branch, before the wait that precedes it in the listing
    hazard.write_text("_Zk:\n" + rd + "\ts_branch .LBB0_2\n.LBB0_1:\n\ts_waitcnt lgkmcnt(0)\n\ts_endpgm\n.LBB0_2:\n\tv_mov_b32_e32 v8, v17\n"
                      "\ts_branch .LBB0_1\n")
    fine = tmp_path / "fine.s"          # the listing shows the copy before the wait; the execution order is wait, then copy
    fine.write_text("_Zk:\n" + rd + "\ts_branch .LBB0_2\n.LBB0_1:\n\tv_mov_b32_e32 v8, v17\n\ts_endpgm\n.LBB0_2:\n\ts_waitcnt lgkmcnt(0)\n"
                    "\ts_branch .LBB0_1\n")
    loop = tmp_path / "loop.s"          # around the back edge: the load of iteration i is consumed in iteration i + 1 after its wait
    loop.write_text("_Zk:\n" + rd + ".LBB0_1:\n\ts_waitcnt lgkmcnt(0)\n\tv_mov_b32_e32 v8, v17\n" + rd +
                    "\ts_cbranch_scc1 .LBB0_1\n\ts_waitcnt lgkmcnt(0)\n\ts_endpgm\n")
    loop_bad = tmp_path / "loop_bad.s"  # the same loop reading v17 once more after the new load was issued
    loop_bad.write_text("_Zk:\n" + rd + ".LBB0_1:\n\ts_waitcnt lgkmcnt(0)\n" + rd + "\tv_mov_b32_e32 v8, v17\n"
                        "\ts_cbranch_scc1 .LBB0_1\n\ts_waitcnt lgkmcnt(0)\n\ts_endpgm\n")
    assert check_asm_inflight.check(str(hazard)) == 1
    assert check_asm_inflight.check(str(fine)) == 0
    assert check_asm_inflight.check(str(loop)) == 0
    assert check_asm_inflight.check(str(loop_bad)) == 1
