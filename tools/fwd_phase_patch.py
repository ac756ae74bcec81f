#!/usr/bin/env python
"""Writes a copy of csrc/isr_forward_fast.hip with cycle counters around the phases of k_render_fwd_fast_w (scan / stage / walk,
s_memtime) into the STATS instance's counters 8..12; build it with tools/build_variant.sh tm isr_api_forward_fast -DISR_TIMING
after swapping the file in, read it with gpurun_in/fwd_timing.py.  Usage: fwd_phase_patch.py apply|revert"""
import os, shutil, sys
P = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "instascene_amd", "csrc", "isr_forward_fast.hip")
B = "/tmp/isr_forward_fast.hip.orig"
if sys.argv[1] == "revert":
    shutil.copy(B, P); sys.exit(0)
shutil.copy(P, B)
s = open(P).read()
i = s.index("k_render_fwd_fast_w(")
head, w = s[:i], s[i:]
def rep(a, b):
    global w
    assert a in w, a
    w = w.replace(a, b, 1)
rep("    int scan = 0, head = 0, pend = 0;                 // wave-uniform\n",
    "    int scan = 0, head = 0, pend = 0;                 // wave-uniform\n"
    "    long long tm_scan = 0, tm_stage = 0, tm_walk = 0, tm_t0 = clock64(), tm_a, tm_b;\n"
    "#define TW_A() tm_a = clock64()\n#define TW_B(acc) do { tm_b = clock64(); acc += tm_b - tm_a; } while (0)\n")
rep("        // ---- scan:", "        TW_A();\n        // ---- scan:")
rep("        if (pend == 0) break;\n        const int nh = min(pend, NH);\n",
    "        TW_B(tm_scan);\n        if (pend == 0) break;\n        const int nh = min(pend, NH);\n        TW_A();\n")
rep("        // ---- walk\n", "        TW_B(tm_stage);\n        TW_A();\n        // ---- walk\n")
rep("        if (m_done == ~0ull) break;\n        head = (head + nh)", "        TW_B(tm_walk);\n        if (m_done == ~0ull) break;\n        head = (head + nh)")
rep("            atomicAdd(stats + 5, (unsigned long long)st_sub);\n",
    "            atomicAdd(stats + 5, (unsigned long long)st_sub);\n"
    "            atomicAdd(stats + 8, (unsigned long long)(clock64() - tm_t0));\n"
    "            atomicAdd(stats + 9, (unsigned long long)tm_scan);\n"
    "            atomicAdd(stats + 10, (unsigned long long)tm_stage);\n"
    "            atomicAdd(stats + 11, (unsigned long long)tm_walk);\n"
    "            atomicAdd(stats + 12, 1ull);\n"
    "            stats[16 + 2 * (size_t)blockIdx.x] = (unsigned long long)tm_t0;\n"
    "            stats[17 + 2 * (size_t)blockIdx.x] = (unsigned long long)clock64();\n")
open(P, "w").write(head + w)
