"""The ISA lint of tools/check_exec_prologue.py (VGPR spill code placed ahead of an exec restore,
the root cause of the wrong Phong pixels at 8 waves per SIMD): it recognises the fault, leaves correct code alone, and the
listing of the library as built is clean."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

FAULTY = """
_ZN2vx4testEv:
	s_and_saveexec_b64 s[6:7], s[4:5]
	s_cbranch_execz .LBB0_2
; %bb.1:
	v_add_f32_e32 v4, v4, v5
.LBB0_2:
	scratch_store_dword off, v58, off offset:40 ; 4-byte Folded Spill
	s_or_b64 exec, exec, s[6:7]
	v_mov_b32_e32 v1, v4
	s_endpgm
"""
CLEAN = """
_ZN2vx4testEv:
	scratch_store_dword off, v58, off offset:40 ; 4-byte Folded Spill
	s_and_saveexec_b64 s[6:7], s[4:5]
	s_cbranch_execz .LBB0_2
; %bb.1:
	v_add_f32_e32 v4, v4, v5
	scratch_store_dword off, v4, off offset:44 ; 4-byte Folded Spill
.LBB0_2:
	s_or_b64 exec, exec, s[6:7]
	scratch_load_dword v58, off, off offset:40 ; 4-byte Folded Reload
	s_endpgm
"""
# the join block printed without a label (short if-body, skip branch dropped)
FAULTY_NO_LABEL = """
_ZN2vx4testEv:
.LBB0_1:
	s_and_saveexec_b64 s[6:7], s[4:5]
	v_add_f32_e32 v4, v4, v5
	scratch_store_dword off, v58, off offset:40 ; 4-byte Folded Spill
	s_or_b64 exec, exec, s[6:7]
	s_endpgm
"""
CLEAN_NO_LABEL = """
_ZN2vx4testEv:
.LBB0_1:
	s_and_saveexec_b64 s[6:7], s[4:5]
	v_add_f32_e32 v4, v4, v5
	scratch_store_dword off, v4, off offset:40 ; 4-byte Folded Spill
	s_or_b64 exec, exec, s[6:7]
	s_endpgm
"""


def _scan(tmp_path, text):
    import check_exec_prologue as L
    f = tmp_path / "k.s"
    f.write_text(text)
    findings, stats = L.scan(str(f))
    return findings


def test_lint_flags_spill_ahead_of_exec_restore(tmp_path):
    f = _scan(tmp_path, FAULTY)
    assert len(f) == 1 and f[0][1] == ".LBB0_2" and "v58" in f[0][2][0][1]
    assert len(_scan(tmp_path, FAULTY_NO_LABEL)) == 1


def test_lint_accepts_correct_placement(tmp_path):
    assert _scan(tmp_path, CLEAN) == []
    assert _scan(tmp_path, CLEAN_NO_LABEL) == []


def test_built_library_listing_is_clean(native_lib):
    """volxel_amd/csrc/Makefile only accepts an object whose listing passes; the listing beside the library says so"""
    listing = os.path.join(ROOT, "volxel_amd", "csrc", "vx_api.s")
    if not os.path.exists(listing):   # a library shipped without its build directory
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "volxel_amd", "csrc"), "-s", "-B", "vx_api.o"])
    import check_exec_prologue as L
    findings, stats = L.scan(listing)
    assert findings == []
    assert sum(1 for k in stats if "render_" in k) >= 20


def test_necessary_instruction_model_matches_the_listing(native_lib):
    """bench.py's roofline fraction multiplies a HAND-DERIVED count -- NECESSARY_VALU: 35 vector instructions per sample, 13
    more per sample inside the sample range; NECESSARY_CLK: the same priced at the measured opcode rates -- by counted samples.
    This ties the hand count to what the compiler actually emitted for the headline kernel's all-lanes march loop
    (tools/isa_cost.py march_loop_counts on the listing that came out of the library's own compile): the loop may not be
    cheaper than the "necessary" count claims (then the model would overstate the floor), and it should not be much dearer
    (then the kernel left the floor and the fraction is stale)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_model", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    import isa_cost
    listing = os.path.join(ROOT, "volxel_amd", "csrc", "vx_api.s")
    if not os.path.exists(listing):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "volxel_amd", "csrc"), "-s", "-B", "vx_api.o"])
    m = isa_cost.march_loop_counts(listing)
    need, clk = b.NECESSARY_VALU, b.NECESSARY_CLK
    assert m["tap_reads"] == 4                                            # eight taps as four ds_read2_b32
    assert need["per_sample"] <= m["per_step"] <= need["per_sample"] + 2, m
    assert clk["per_sample"] <= m["per_step_clk"] <= clk["per_sample"] + 8, m
    # the composite block also holds what the model does not call necessary: the upper range compare, the alpha select,
    # the termination select and one register move
    assert need["per_tf_sample"] <= m["per_tf_step"] <= need["per_tf_sample"] + 5, m
    assert clk["per_tf_sample"] <= m["per_tf_step_clk"] <= clk["per_tf_sample"] + 16, m
