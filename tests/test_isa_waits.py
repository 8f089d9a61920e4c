"""The bf16 weight-gradient kernel overlaps the LDS-DMA of K-tile k+1 with the products of K-tile k only as
long as the compiler does not order the transposing LDS reads behind that transfer: its wait-count pass
does so (s_waitcnt vmcnt(0) right behind the buffer_load ... lds instructions) whenever it cannot tell the
two LDS buffers apart.  This compiles the kernel for gfx950 and reads the loop: a property of the generated
code that no numerical test can see (the results are the same, the kernel is 15 % slower)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def loop_events(body):
    """the K loop (first inner loop of the kernel) as a sequence of events"""
    a = body.index("Inner Loop Header")
    b = body.index("s_cbranch_scc1", a)
    seq = []
    for line in body[a:b].split("\n"):
        line = line.strip()
        if line.startswith("buffer_load") and line.endswith("lds"):
            seq.append("dma")
        elif "s_waitcnt" in line and "vmcnt(0)" in line:
            seq.append("wait_all")
        elif line.startswith("ds_read_b64_tr_b16"):
            seq.append("read")
        elif line.startswith("s_barrier"):
            seq.append("barrier")
    return seq


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_weight_gradient_loop_does_not_wait_for_the_tile_it_has_just_asked_for(tmp_path):
    src = os.path.join(ROOT, "nerf_amd", "csrc", "gemm_bf16.hip")
    out = tmp_path / "gemm_bf16.s"
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-S", "--cuda-device-only",
                    "-o", str(out), src], check=True, capture_output=True, timeout=600)
    text = out.read_text()
    kernels = re.findall(r"^(\w*wgrad_bf16_kernel\w*):", text, re.M)
    assert len(kernels) == 2, kernels
    for name in kernels:
        start = text.index(name + ":")
        seq = loop_events(text[start:text.index("s_endpgm", start)])
        assert "dma" in seq and "read" in seq and "barrier" in seq, name
        for i, ev in enumerate(seq):
            if ev != "wait_all":
                continue
            # a full wait belongs in front of the barrier that ends a K-tile, never between a transfer and reads
            rest = [e for e in seq[i + 1:] if e != "wait_all"]
            assert rest and rest[0] == "barrier", (name, seq[max(0, i - 3):i + 4])
