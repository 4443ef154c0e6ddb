#!/usr/bin/env python
"""Evaluates away the `#if ... EXP_* ...` timing-experiment blocks of a HIP source (every EXP_* macro undefined) and writes the
result back.  Round 2 kept ~35 such blocks ("numerically wrong on purpose") inside the product kernels; since round 3 the product
translation units carry none -- the experiment variants live in git history (tools/exp/README.md says where and how to rebuild
them).  Usage: python tools/strip_exp.py file.hip [...]"""
import re
import sys

MACRO = r"(?:EXP_\w+|W_DMA_IN_PHASE|W_BARRIER_EVERY_CHUNK)"    # the switches (and the two names round 2 derived from them)


def truth(expr):
    """value of a preprocessor condition made of EXP_* macros only (all undefined), or None if it mentions anything else"""
    e = expr.strip()
    e = re.sub(r"defined\s*\(\s*" + MACRO + r"\s*\)", "0", e)
    e = re.sub(r"defined\s+" + MACRO, "0", e)
    if re.search(r"[A-Za-z_]", e):
        return None
    e = e.replace("||", " or ").replace("&&", " and ").replace("!", " not ")
    return bool(eval(e))


def strip(text):
    out, stack = [], []     # stack entries: dict(exp, taken, active, parent_active)
    active = lambda: all(s["active"] for s in stack)
    for line in text.split("\n"):
        m = re.match(r"\s*#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)", line)
        if not m:
            if active():
                out.append(line)
            continue
        kw, rest = m.group(1), m.group(2).split("//")[0].strip()
        if kw in ("ifdef", "ifndef", "if"):
            if kw == "if":
                val = truth(rest)
            else:
                val = (kw == "ifndef") if re.fullmatch(MACRO, rest) else None
            if val is None:                       # not an experiment switch: keep verbatim
                stack.append(dict(exp=False, active=True, taken=True))
                if all(s["active"] for s in stack[:-1]):
                    out.append(line)
            else:
                stack.append(dict(exp=True, active=val, taken=val))
        elif kw == "elif":
            s = stack[-1]
            if not s["exp"]:
                if active():
                    out.append(line)
                continue
            val = truth(rest)
            if val is None:
                raise SystemExit("mixed #elif chain not supported: " + line)
            s["active"] = (not s["taken"]) and val
            s["taken"] = s["taken"] or val
        elif kw == "else":
            s = stack[-1]
            if not s["exp"]:
                if active():
                    out.append(line)
                continue
            s["active"] = not s["taken"]
            s["taken"] = True
        else:
            s = stack.pop()
            if not s["exp"] and active():
                out.append(line)
    assert not stack
    return "\n".join(out)


if __name__ == "__main__":
    for path in sys.argv[1:]:
        src = open(path).read()
        new = strip(src)
        open(path, "w").write(new)
        print(path, src.count("\n") - new.count("\n"), "lines removed; EXP_ left:", new.count("EXP_"))
