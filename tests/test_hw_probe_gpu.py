"""Builds tests/hw_probe/probe.hip on the GPU box and asserts the two gfx950 hardware behaviours the contraction and
attention kernels rely on (no prebuilt binary is kept in the tree):
  * an LDS-DMA lane whose buffer offset is out of range writes ZEROS into LDS (how padding / ragged tiles are staged);
  * ds_read_b64_tr_b16: lane i of a 16-lane group receives column i of the group's 4 rows (the free transpose the
    mn-major operands and the attention V panels use)."""
import os
import re
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def test_lds_dma_zero_fill_and_tr16_lane_map(cuda, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "probe")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", os.path.join(HERE, "hw_probe", "probe.hip"), "-o", exe])
    out = subprocess.run([exe], check=True, capture_output=True, text=True, timeout=120).stdout
    lanes = dict(re.findall(r"lane(\d+):([0-9a-f]{8})", out.split("second DMA")[0]))
    for l in range(8):
        want = 0 if l & 1 else 0x1000 + 4 * l          # odd lanes: out-of-range offset -> zeros; even: their 16-byte chunk
        assert int(lanes[str(l)], 16) == want, (l, lanes[str(l)])
    # (the probe's second DMA uses an instruction offset, which the kernels never do: printed for the record, not asserted)
    got = {int(l): re.findall(r"\((\d+),(\d+)\)", body) for l, body in re.findall(r"L(\d\d):((?:\(\d+,\d+\))+)", out)}
    assert len(got) == 64
    for l in range(64):
        g, i = l >> 4, l & 15
        assert [(int(r), int(c)) for r, c in got[l]] == [(4 * g + j, i) for j in range(4)], (l, got[l])
