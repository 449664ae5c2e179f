"""tools/fypp_lite.py: the subset of the Fypp language that the reference's Fortran sources use (the expander that lets the UNCHANGED
reference build here).  Expected outputs follow the Fypp manual's semantics."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import fypp_lite as F  # noqa: E402


def expand(text, tmp_path, name="t.F", extra=None):
    for fn, body in (extra or {}).items():
        (tmp_path / fn).write_text(body)
    p = tmp_path / name
    p.write_text(text)
    return F.Expander(include_dirs=[str(tmp_path)]).expand_file(str(p))


def test_set_inline_and_line_evaluation(tmp_path):
    out = expand("#:set n = 3\n#:set names = ['a', 'b']\nx = ${n * 2}$ ! ${names[1]}$\n$: 'y = %d' % (n + 1)\n", tmp_path)
    assert out == "x = 6 ! b\ny = 4\n"


def test_for_if_elif_else(tmp_path):
    src = ("#:for t, k in [('real', 8), ('real', 4), ('complex', 8)]\n"
           "#:if t == 'real' and k == 8\nd\n#:elif t == 'real'\ns\n#:else\nz\n#:endif\n"
           "#:endfor\n")
    assert expand(src, tmp_path) == "d\ns\nz\n"


def test_def_call_forms_and_local_set(tmp_path):
    src = ("#:set who = 'outer'\n"
           "#:def greet(x, y='!')\n#:set who = 'inner'\nhello ${x}$${y}$ from ${who}$\n#:enddef\n"
           "$: greet('a')\n"
           "$: greet('b', y='?')\n"
           "call @{greet(c)}@\n"           # direct call: the argument is TEXT, not a Python expression
           "${who}$\n")
    assert expand(src, tmp_path) == "hello a! from inner\nhello b? from inner\ncall hello c! from inner\nouter\n"


def test_mute_keeps_definitions_and_comments_vanish(tmp_path):
    src = "#:mute\n#:set v = 5\nthis text disappears\n#:endmute\n#! a preprocessor comment\nv = ${v}$\n"
    assert expand(src, tmp_path) == "v = 5\n"


def test_include_and_continuation(tmp_path):
    inc = "#:set kinds = ['s', 'd']\n#:def suffix(k)\n_${k}$\n#:enddef\n"
    src = ('#:include "defs.fypp"\n'
           "#:for k in &\n     & kinds\nsub${suffix(k)}$\n#:endfor\n")
    assert expand(src, tmp_path, extra={"defs.fypp": inc}) == "sub_s\nsub_d\n"


def test_comprehension_sees_the_enclosing_scope(tmp_path):
    # (the reference builds its type lists this way, e.g. src/data/dbcsr_data_methods_low.F)
    src = "#:set base = ['r', 'c']\n#:set sizes = [4, 8]\n#:set inst = [b + str(s) for b in base for s in sizes]\n${', '.join(inst)}$\n"
    assert expand(src, tmp_path) == "r4, r8, c4, c8\n"


def test_unknown_directive_is_an_error(tmp_path):
    with pytest.raises(F.FyppError):
        expand("#:frobnicate x\n", tmp_path)
