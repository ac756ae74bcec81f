#!/usr/bin/env python
"""Swaps in copies of csrc/isr_forward_fast.hip / isr_api_forward_fast.hip in which the PRODUCTION instance of k_render_fwd_fast_w
(no work counters, no atomics) stamps its phases with s_memtime and stores, per wave and with plain stores,
(total, scan, stage, walk, splats walked) at counters[8 * workgroup ..]; build with tools/build_variant.sh tm
isr_api_forward_fast, read with gpurun_in/fwd_timing.py.  Usage: fwd_phase_patch.py apply|revert"""
import os, shutil, sys
D = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "instascene_amd", "csrc")
P, A = os.path.join(D, "isr_forward_fast.hip"), os.path.join(D, "isr_api_forward_fast.hip")
if sys.argv[1] == "revert":
    shutil.copy("/tmp/isr_forward_fast.hip.orig", P); shutil.copy("/tmp/isr_api_forward_fast.hip.orig", A); sys.exit(0)
shutil.copy(P, "/tmp/isr_forward_fast.hip.orig"); shutil.copy(A, "/tmp/isr_api_forward_fast.hip.orig")
s = open(P).read()
i = s.index("k_render_fwd_fast_w(")
head, w = s[:i], s[i:]
def rep(a, b):
    global w
    assert a in w, a
    w = w.replace(a, b, 1)
rep("    int scan = 0, head = 0, pend = 0;                 // wave-uniform\n",
    "    int scan = 0, head = 0, pend = 0;                 // wave-uniform\n"
    "    long long tm_scan = 0, tm_stage = 0, tm_walk = 0, tm_t0 = clock64(), tm_a, tm_b; unsigned tm_n = 0;\n"
    "#define TW_A() tm_a = clock64()\n#define TW_B(acc) do { tm_b = clock64(); acc += tm_b - tm_a; } while (0)\n")
rep("        // ---- scan:", "        TW_A();\n        // ---- scan:")
rep("        if (pend == 0) break;\n        const int nh = min(pend, NH);\n",
    "        TW_B(tm_scan);\n        if (pend == 0) break;\n        const int nh = min(pend, NH);\n        TW_A();\n")
rep("        // ---- walk\n", "        TW_B(tm_stage);\n        TW_A();\n        // ---- walk\n")
rep("            if (STATS) st_eval++;\n", "            if (STATS) st_eval++;\n            tm_n++;\n")
rep("        if (m_done == ~0ull) break;\n        head = (head + nh)", "        TW_B(tm_walk);\n        if (m_done == ~0ull) break;\n        head = (head + nh)")
rep("    if (STATS) {\n        if (lane == 0 && first_pass) {",
    "    if (!STATS && stats != nullptr && lane == 0 && first_pass) {\n"
    "        unsigned long long* o = stats + 8 * (size_t)blockIdx.x;\n"
    "        o[0] = (unsigned long long)(clock64() - tm_t0); o[1] = (unsigned long long)tm_scan; o[2] = (unsigned long long)tm_stage;\n"
    "        o[3] = (unsigned long long)tm_walk; o[4] = tm_n;\n    }\n"
    "    if (STATS) {\n        if (lane == 0 && first_pass) {")
open(P, "w").write(head + w)
a = open(A).read()
n0 = a.count("if (counters) ISR_GW(")
a = a.replace("if (counters) ISR_GW(false, true, 1); else ISR_GW(false, false, 1);", "ISR_GW(false, false, 1);")
a = a.replace("if (counters) ISR_GW(true, true, 2); else ISR_GW(true, false, 2);", "ISR_GW(true, false, 2);")
a = a.replace("if (counters) ISR_GW(true, true, 1); else ISR_GW(true, false, 1);", "ISR_GW(true, false, 1);")
assert a.count("if (counters) ISR_GW(") == 0 and n0 == 3
open(A, "w").write(a)
