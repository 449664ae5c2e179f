#!/usr/bin/env python3
"""fypp_lite -- expander for the subset of the Fypp preprocessor language that the
DBCSR Fortran sources use.

The reference needs the `fypp` tool for 39 of its 96 Fortran sources
(/root/reference/cmake/fypp-sources.cmake:1-12); the tool lives in an empty git
submodule and cannot be fetched.  This is an independent implementation of the
directives that actually occur in /root/reference/src (counted with grep):

    #:set  #:for/#:endfor  #:if/#:elif/#:else/#:endif  #:def/#:enddef
    #:include  #:mute/#:endmute  #! comments  $: line evaluation
    ${ expr }$ inline evaluation, @{ macro(text args) }@ inline direct call,
    '&' continuation of directive lines

Semantics follow the Fypp manual: expressions are Python expressions evaluated in
the current scope; a macro (#:def) is a callable that returns its rendered body
without the final newline; macro bodies see their arguments, then the scope of the
definition; #:set inside a macro is local; #:mute discards rendered text but keeps
definitions; #:include is relative to the including file.  Lines are NOT folded at
132 columns (flang accepts long free-form lines).

Usage: fypp_lite.py [-I dir]... input output     (build-container tool; expands the
reference where it lies, output goes to a scratch directory, never into the repo).
"""
import os
import re
import sys

_DIR = re.compile(r"^\s*#:\s*(\w+)\s*(.*)$")
_COMMENT = re.compile(r"^\s*#!")
_LINE_EVAL = re.compile(r"^\s*\$:\s?(.*)$")
_INLINE = re.compile(r"\$\{(.*?)\}\$")
_DIRECT = re.compile(r"@\{(\w+)\((.*?)\)\}@")


class FyppError(Exception):
    pass


class Scope(dict):
    """dict with a parent chain; used as the `locals` mapping of eval()."""

    def __init__(self, parent=None):
        super().__init__()
        self.parent = parent

    def __missing__(self, key):
        p = self.parent
        while p is not None:
            if dict.__contains__(p, key):
                return dict.__getitem__(p, key)
            p = p.parent
        raise KeyError(key)

    def __contains__(self, key):
        try:
            self[key]
            return True
        except KeyError:
            return False


def _join_continuations(lines):
    """directive / line-eval lines ending in '&' continue on the next line (optional leading '&')."""
    out, i = [], 0
    while i < len(lines):
        ln = lines[i]
        if (_DIR.match(ln) or _LINE_EVAL.match(ln)) and ln.rstrip().endswith("&"):
            acc = ln.rstrip()[:-1]
            n = 1
            while i + n < len(lines):
                nxt = lines[i + n].strip()
                if nxt.startswith("&"):
                    nxt = nxt[1:]
                n += 1
                if nxt.endswith("&"):
                    acc += nxt[:-1]
                else:
                    acc += nxt
                    break
            out.append(acc)
            out.extend([None] * (n - 1))  # keep line numbering for messages
            i += n
        else:
            out.append(ln)
            i += 1
    return out


def parse(lines, fname):
    """-> nested node list: ('text', str) | ('eval', expr) | ('set', name, expr) | ('for', vars, expr, body)
       | ('if', [(cond|None, body)...]) | ('def', name, argspec, body) | ('include', path) | ('mute', body)"""
    lines = _join_continuations(lines)
    pos = 0

    def block(terminators):
        nonlocal pos
        nodes = []
        while pos < len(lines):
            ln = lines[pos]
            lineno = pos + 1
            pos += 1
            if ln is None or _COMMENT.match(ln):
                continue
            m = _DIR.match(ln)
            if not m:
                me = _LINE_EVAL.match(ln)
                nodes.append(("eval", me.group(1).strip()) if me else ("text", ln))
                continue
            kw, rest = m.group(1), m.group(2).strip()
            if kw in terminators:
                return nodes, kw, rest
            if kw == "set":
                if "=" in rest:
                    name, expr = rest.split("=", 1)
                    nodes.append(("set", name.strip(), expr.strip()))
                else:
                    nodes.append(("set", rest, "True"))
            elif kw == "for":
                mm = re.match(r"(.+?)\s+in\s+(.+)$", rest)
                if not mm:
                    raise FyppError("%s:%d: malformed #:for" % (fname, lineno))
                body, end, _ = block(("endfor",))
                nodes.append(("for", [v.strip() for v in mm.group(1).split(",")], mm.group(2), body))
            elif kw == "if":
                branches, cond = [], rest
                while True:
                    body, end, erest = block(("elif", "else", "endif"))
                    branches.append((cond, body))
                    if end == "endif":
                        break
                    cond = erest if end == "elif" else None
                nodes.append(("if", branches))
            elif kw == "def":
                mm = re.match(r"(\w+)\s*\((.*)\)\s*$", rest)
                if not mm:
                    raise FyppError("%s:%d: malformed #:def" % (fname, lineno))
                body, _, _ = block(("enddef",))
                nodes.append(("def", mm.group(1), mm.group(2), body))
            elif kw == "mute":
                body, _, _ = block(("endmute",))
                nodes.append(("mute", body))
            elif kw == "include":
                nodes.append(("include", rest.strip().strip("'\"")))
            else:
                raise FyppError("%s:%d: unsupported directive #:%s" % (fname, lineno, kw))
        if terminators:
            raise FyppError("%s: missing %s" % (fname, "/".join(terminators)))
        return nodes, None, None

    nodes, _, _ = block(())
    return nodes


class Expander:
    def __init__(self, include_dirs=()):
        self.include_dirs = list(include_dirs)
        self.globals = Scope()
        self.pyglobals = {"__builtins__": __builtins__}

    @staticmethod
    def flat(scope):
        """one plain dict for eval(): comprehensions and lambdas inside an expression only see eval's GLOBALS"""
        chain, d = [], {"__builtins__": __builtins__}
        while scope is not None:
            chain.append(scope)
            scope = scope.parent
        for sc in reversed(chain):
            d.update(sc)
        return d

    def ev(self, expr, scope, where):
        try:
            return eval(expr, self.flat(scope))
        except Exception as e:  # noqa: BLE001
            raise FyppError("%s: cannot evaluate %r: %s" % (where, expr, e))

    @staticmethod
    def split_args(text):
        """arguments of a direct call @{name(a, b(c, d))}@: raw text, split at top-level commas"""
        args, depth, cur = [], 0, ""
        for ch in text:
            if ch == "," and depth == 0:
                args.append(cur.strip())
                cur = ""
                continue
            depth += ch in "([" 
            depth -= ch in ")]"
            cur += ch
        if cur.strip() or args:
            args.append(cur.strip())
        return args

    def subst(self, text, scope, where):
        text = _INLINE.sub(lambda m: str(self.ev(m.group(1).strip(), scope, where)), text)
        if "@{" in text:  # direct call: the arguments are passed as TEXT, not evaluated
            def call(m):
                fn = self.ev(m.group(1), scope, where)
                return str(fn(*self.split_args(m.group(2))))
            text = _DIRECT.sub(call, text)
        return text

    def render(self, nodes, scope, fname):
        out = []
        for nd in nodes:
            k = nd[0]
            if k == "text":
                out.append(self.subst(nd[1], scope, fname))
            elif k == "eval":
                v = self.ev(self.subst(nd[1], scope, fname), scope, fname)
                out.append(("" if v is None else str(v)) + "\n")
            elif k == "set":
                val = self.ev(nd[2], scope, fname)
                names = [n.strip() for n in nd[1].strip("()").split(",")]
                if len(names) == 1:
                    scope[names[0]] = val
                else:
                    for n, v in zip(names, val):
                        scope[n] = v
            elif k == "for":
                for item in self.ev(nd[2], scope, fname):
                    if len(nd[1]) == 1:
                        scope[nd[1][0]] = item
                    else:
                        for n, v in zip(nd[1], item):
                            scope[n] = v
                    out.append(self.render(nd[3], scope, fname))
            elif k == "if":
                for cond, body in nd[1]:
                    if cond is None or self.ev(cond, scope, fname):
                        out.append(self.render(body, scope, fname))
                        break
            elif k == "def":
                scope[nd[1]] = self.make_macro(nd[1], nd[2], nd[3], scope, fname)
            elif k == "mute":
                self.render(nd[1], scope, fname)
            elif k == "include":
                out.append(self.expand_file(self.find(nd[1], fname), scope))
        return "".join(out)

    def make_macro(self, name, argspec, body, defscope, fname):
        exp = self
        # bind positional / keyword arguments with Python's own rules (defaults are evaluated at definition time, as in Fypp)
        binder = eval("lambda %s: locals()" % argspec, self.flat(defscope))

        def macro(*args, **kwargs):
            local = Scope(defscope)
            local.update(binder(*args, **kwargs))
            text = exp.render(body, local, "%s (macro %s)" % (fname, name))
            return text[:-1] if text.endswith("\n") else text

        macro.__name__ = name
        return macro

    def find(self, inc, fname):
        for d in [os.path.dirname(os.path.abspath(fname))] + self.include_dirs:
            p = os.path.join(d, inc)
            if os.path.exists(p):
                return p
        raise FyppError("%s: include file %r not found" % (fname, inc))

    def expand_file(self, path, scope=None):
        with open(path) as f:
            lines = f.read().splitlines(keepends=True)
        if lines and not lines[-1].endswith("\n"):
            lines[-1] += "\n"
        return self.render(parse(lines, path), self.globals if scope is None else scope, path)


def main(argv):
    incs, args = [], []
    i = 1
    while i < len(argv):
        if argv[i] == "-I":
            incs.append(argv[i + 1])
            i += 2
        elif argv[i].startswith("-I"):
            incs.append(argv[i][2:])
            i += 1
        else:
            args.append(argv[i])
            i += 1
    if len(args) != 2:
        sys.stderr.write(__doc__)
        return 2
    text = Expander(incs).expand_file(args[0])
    os.makedirs(os.path.dirname(os.path.abspath(args[1])), exist_ok=True)
    with open(args[1], "w") as f:
        f.write(text)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
