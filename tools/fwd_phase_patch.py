#!/usr/bin/env python
"""Swaps in a copy of csrc/isr_forward_fast.hip with cycle counters (s_memtime) around the phases of k_render_fwd_fast_w
(DMA wait / convert / scan + issue / walk) in the STATS instance's counters 8..14; build with
tools/build_variant.sh tm isr_api_forward_fast, read with gpurun_in/fwd_timing.py.  Usage: fwd_phase_patch.py apply|revert"""
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
rep("    int scan = 0, head = 0, pend = 0, buf = 0;                 // wave-uniform\n",
    "    int scan = 0, head = 0, pend = 0, buf = 0;                 // wave-uniform\n"
    "    long long tm_issue = 0, tm_wait = 0, tm_conv = 0, tm_scan = 0, tm_walk = 0, tm_t0 = clock64(), tm_a, tm_b;\n"
    "#define TW_A() tm_a = clock64()\n#define TW_B(acc) do { tm_b = clock64(); acc += tm_b - tm_a; } while (0)\n")
rep("        if (!primed) issue(head, nh, buf);\n", "        TW_A();\n        if (!primed) issue(head, nh, buf);\n")
rep("        // ---- convert: lane = hit", "        TW_B(tm_wait); TW_A();\n        // ---- convert: lane = hit")
rep("        // ---- refill the ring", "        asm volatile(\"s_waitcnt lgkmcnt(0)\");\n        TW_B(tm_conv); TW_A();\n        // ---- refill the ring")
rep("        primed = pend > 0;\n", "        TW_B(tm_scan); TW_A();\n        primed = pend > 0;\n")
rep("        // ---- walk\n", "        TW_B(tm_issue); TW_A();\n        // ---- walk\n")
rep("        if (m_done == ~0ull) break;\n        buf ^= 1;", "        TW_B(tm_walk);\n        if (m_done == ~0ull) break;\n        buf ^= 1;")
rep("            atomicAdd(stats + 5, (unsigned long long)st_sub);\n",
    "            atomicAdd(stats + 5, (unsigned long long)st_sub);\n"
    "            atomicAdd(stats + 8, (unsigned long long)(clock64() - tm_t0));\n"
    "            atomicAdd(stats + 9, (unsigned long long)tm_wait);\n"
    "            atomicAdd(stats + 10, (unsigned long long)tm_conv);\n"
    "            atomicAdd(stats + 11, (unsigned long long)tm_walk);\n"
    "            atomicAdd(stats + 12, 1ull);\n"
    "            atomicAdd(stats + 13, (unsigned long long)tm_scan);\n"
    "            atomicAdd(stats + 14, (unsigned long long)tm_issue);\n")
open(P, "w").write(head + w)
