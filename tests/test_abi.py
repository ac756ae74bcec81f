"""The C ABI as a COMPILE-TIME fact: every ctypes signature of instascene_amd/_lib.py (arity, the width and signedness of every
integer, float against double, pointer against value, the return type) is static_assert-ed against the declaration in
include/instascene_rasterizer.h / include/instascene_ops.h by a generated translation unit compiled with g++ (no GPU, no HIP)."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _canon(t):
    if t is None:
        return "void"
    if t in (ctypes.c_void_p, ctypes.c_char_p):
        return "void*"
    if t is ctypes.c_float:
        return "float"
    if t is ctypes.c_double:
        return "double"
    signed = t(-1).value < 0
    return "Int<%d, %s>" % (ctypes.sizeof(t), "true" if signed else "false")


PRELUDE = r"""
#include <type_traits>
#include "instascene_rasterizer.h"
#include "instascene_ops.h"
template <int N, bool S> struct Int {};
template <class... T> struct List {};
template <class T> struct canon {
    using type = std::conditional_t<std::is_pointer_v<T>, void*,
                 std::conditional_t<std::is_integral_v<T>, Int<(int)sizeof(std::conditional_t<std::is_void_v<T>, char, T>), std::is_signed_v<T>>, T>>;
};
template <class F> struct fsig;
template <class R, class... A> struct fsig<R (*)(A...)> { using type = List<typename canon<R>::type, typename canon<A>::type...>; };
"""


def test_ctypes_signatures_match_the_headers(tmp_path):
    sys.path.insert(0, ROOT)
    from instascene_amd import _lib
    lines = [PRELUDE]
    for name, (res, args) in _lib.SIGNATURES.items():
        want = ", ".join([_canon(res)] + [_canon(a) for a in args])
        lines.append('static_assert(std::is_same_v<fsig<decltype(&%s)>::type, List<%s>>, "%s: ctypes signature differs from the header");'
                     % (name, want, name))
    lines.append("int main() { return 0; }")
    src = tmp_path / "abi_check.cpp"
    src.write_text("\n".join(lines))
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_a_wrong_signature_is_caught(tmp_path):
    """The check bites: one argument narrowed from int64_t to int must fail to compile."""
    src = tmp_path / "abi_bad.cpp"
    src.write_text(PRELUDE + 'static_assert(std::is_same_v<fsig<decltype(&isr_binning_bytes)>::type, '
                   'List<Int<8, false>, Int<4, true>, Int<4, true>, Int<4, true>>>, "narrowed");\nint main() { return 0; }\n')
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "narrowed" in r.stderr
