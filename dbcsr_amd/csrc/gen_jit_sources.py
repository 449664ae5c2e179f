#!/usr/bin/env python3
"""Build step: the texts of the headers that hiprtc needs, as C++ raw string literals (jit_sources.inc, not tracked)."""
import sys

out = open(sys.argv[1], "w")
for name in sys.argv[2:]:
    text = open(name).read()
    assert ')DBCSRJIT"' not in text
    out.write('static const char* const kJitSrc_%s = R"DBCSRJIT(%s)DBCSRJIT";\n' % (name.rsplit("/", 1)[-1].replace(".h", ""), text))
out.close()
