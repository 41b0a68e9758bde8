#!/usr/bin/env python3
"""Verify on the generated gfx950 ISA that the fixed VGPRs the k=27 kernels use for in-flight vector-memory data
(v116..v127, see vgmi_kernels.hip) are touched only by the hand-written (VGHOT-tagged) instructions, and that the
compiler's own allocation stays below them.  Usage: check_hot_vgprs.py [path/to/vgmi_kernels.hip]"""
import os, re, subprocess, sys, tempfile

def main():
    src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "varigraph_amd", "csrc", "vgmi_kernels.hip")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                               "-Wno-unused-function", os.path.abspath(src), "-o", out])
        text = open(out).read()
    bad = 0
    found = 0
    for m in re.finditer(r"^(_ZN3vgk14count27_kernelILb[01]ELb[01]EEEvNS_9RowParamsE|_ZN3vgk15count27s_kernelILb[01]ELj\d+EEEvNS_9RowParamsE):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M):
        found += 1
        name, body = m.group(1), m.group(2)
        top = 0
        for ln in body.split("\n"):
            code = ln.split(";")[0] if "VGHOT" not in ln else ""
            for r in re.finditer(r"\bv(\d+)\b", code):
                top = max(top, int(r.group(1)))
            for r in re.finditer(r"\bv\[(\d+):(\d+)\]", code):
                top = max(top, int(r.group(2)))
        n_hot = body.count("VGHOT")
        n_scratch = len(re.findall(r"\bscratch_(load|store)", body))   # stack traffic would also break the vmcnt bookkeeping
        print(f"{name}: {n_hot} hand-written instructions, compiler's highest VGPR v{top}, {n_scratch} scratch accesses")
        limit = 76 if "count27s_kernelILb1" in name else 116     # the path-table variant also hand-manages v76..v115
        if top >= limit or n_hot == 0 or n_scratch:
            bad += 1
    if found != 12 or bad:      # count27_kernel x 3, count27s_kernel<false / true, 27>, count27s_kernel<true, 19 .. 25>
        print("FAILED")
        return 1
    print("OK")
    return 0

if __name__ == "__main__":
    sys.exit(main())
