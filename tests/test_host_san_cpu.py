"""Host sanitizer leg (SURVEY.md §5 'race detection / sanitizers'): the launch planning of the contraction kernel
(multimodal_sc_amd/csrc/gemm_plan.h: tile choice, split-K plans, workspace-size queries, workgroup-id mappings) is built
with -fsanitize=address,undefined and fuzzed against the extents the launches index (tests/host_san/plan_fuzz.cpp).
CPU build only — GPU sanitizers are not available on this pool; the device side is covered by the bitwise-determinism
tests."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_gemm_plan_fuzz_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "plan_fuzz")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-fno-omit-frame-pointer", os.path.join(HERE, "host_san", "plan_fuzz.cpp"), "-o", exe])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    for knob in ({}, {"MSC_CONV_SPLITK": "0"}, {"MSC_WGRAD_BLOCKS": "1536"}, {"MSC_GEMM_TILE": "256x256"}, {"MSC_GEMM_TILE": "64x64"}):
        r = subprocess.run([exe, "6000"], env=dict(env, **knob), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (knob, r.stdout[-2000:], r.stderr[-4000:])
        assert "iterations ok" in r.stdout
