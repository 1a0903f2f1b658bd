#!/usr/bin/env python3
"""ISA check behind DESIGN.md §3.1's rule: with LDS-DMA in flight, the compiler may place `s_waitcnt vmcnt(0)` in front of an LDS read it cannot prove
independent of the DMA writes (seen for the ds_read_b64_tr_b16 builtin: the next stage's prefetch then never overlaps the MFMAs).  Compiles the given
csrc/*.hip files to gfx950 assembly (device only, ~2 min for gemm.hip) and reports every `s_waitcnt vmcnt(0)` that sits between an LDS-DMA instruction
(buffer_load ... lds, at most WINDOW lines above, no s_barrier in between) and an LDS read (at most 6 lines below).  Exit code 1 if any is found.
    usage: tools/isa_dma_wait_check.py [csrc/gemm.hip csrc/conv_c64w.hip csrc/conv_stem.hip ...]   (default: those three)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "multimodal_sc_amd", "csrc")
WINDOW = 60
files = sys.argv[1:] or ["gemm.hip", "conv_c64w.hip", "conv_stem.hip"]
bad = 0
for f in files:
    src = f if os.path.isabs(f) else os.path.join(CSRC, os.path.basename(f))
    with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-result", "-I" + os.path.join(ROOT, "include"), "-S",
                        "--cuda-device-only", src, "-o", tmp.name], check=True, cwd=CSRC, stderr=subprocess.DEVNULL)
        lines = open(tmp.name).read().split("\n")
    kernel = "?"
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            kernel = m.group(1)
        if not re.search(r"s_waitcnt vmcnt\(0\)(?! lgkmcnt)", l) and "s_waitcnt vmcnt(0)" not in l:
            continue
        below = lines[i + 1:i + 7]
        if not any(re.search(r"\bds_read", b) for b in below) or any("s_barrier" in b for b in below):
            continue                       # a wait in front of a barrier is the source's own stage wait
        dma = None
        for k in range(i - 1, max(i - WINDOW, 0), -1):
            if "s_barrier" in lines[k]:
                break
            if re.search(r"buffer_load_dword.* lds", lines[k]):
                dma = k
                break
        if dma is not None:
            bad += 1
            print(f"{os.path.basename(src)}: {kernel[:90]}: line {i + 1}: vmcnt(0) {i - dma} lines behind an LDS-DMA issue, in front of {[b.strip().split()[0] for b in below if 'ds_read' in b][0]}")
print(f"{bad} compiler-placed DMA waits in front of LDS reads" if bad else "clean: no vmcnt(0) between an LDS-DMA issue and the LDS reads that follow it")
sys.exit(1 if bad else 0)
