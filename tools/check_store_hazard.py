#!/usr/bin/env python3
"""Scan the built library's gfx950 code for the 16-byte buffer-store hazard (ubresnet_amd/csrc/ubr_conv.hip, buf_store16).

A `buffer_store_dwordx3/x4` whose soffset is an SGPR gets no wait state from hipcc (ROCm 7.2) before a VALU write to its data
registers; on gfx950 that corrupts the stored data.  The sources avoid the form (scalar offsets are folded into the vector offset
for 16-byte stores); this check makes sure a later edit does not bring it back.

usage: python tools/check_store_hazard.py [path/to/lib.so]   -> exit code 1 when a store of that form is found"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
STORE = re.compile(r"\bbuffer_store_dwordx[34]\s+v\[(\d+):(\d+)\],\s*(\S+),\s*s\[\d+:\d+\],\s*(\S+)")


def scan(lib):
    """-> (number of 12/16-byte buffer stores, [offending lines])"""
    tmp = tempfile.mkdtemp(prefix="ubr_hz_")
    try:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(lib, so)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        total, bad = 0, []
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
            for line in dis.split("\n"):
                m = STORE.search(line)
                if not m:
                    continue
                total += 1
                if re.fullmatch(r"s\d+|s\[\d+:\d+\]|m0|vcc_lo|vcc_hi|ttmp\d+", m.group(4)):
                    bad.append(line.strip())
        return total, bad
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "ubresnet_amd", "libubresnet_hip.so")
    n, bad = scan(lib)
    print("%d buffer stores of 12/16 bytes, %d with an SGPR soffset" % (n, len(bad)))
    for b in bad[:20]:
        print("  ", b)
    sys.exit(1 if bad else 0)
