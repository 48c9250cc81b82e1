#!/usr/bin/env python3
"""Static check of integration/*.go against the C ABI: every C.<function> the Go side calls must be declared in
include/ykpred.h or include/ykhost.h (with the same number of arguments), every C.<CONSTANT> must be #defined there, and
every C.<type> must be a typedef / struct of the headers. There is no Go toolchain in the build image, so this is what keeps
the Go source honest. Exit code 0 = consistent; prints what it checked."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBC = {"CString", "GoString", "free", "malloc", "calloc", "char", "int32_t", "uint32_t", "int64_t", "uint8_t", "size_t"}


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def split_args(arglist):
    depth, cur, out = 0, "", []
    for ch in arglist:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def call_args(text, start):
    """Argument string of the call whose '(' is at text[start]."""
    depth = 0
    for i in range(start, len(text)):
        if text[i] == "(":
            depth += 1
        elif text[i] == ")":
            depth -= 1
            if depth == 0:
                return text[start + 1:i]
    raise ValueError("unbalanced call")


def main():
    headers = ""
    for h in ("ykpred.h", "ykhost.h"):
        headers += strip_comments(open(os.path.join(ROOT, "include", h)).read()) + "\n"
    functions = {}
    for m in re.finditer(r"\b(yk(?:pred|host)_\w+)\s*\(", headers):
        args = call_args(headers, m.end() - 1)
        functions[m.group(1)] = 0 if args.strip() in ("", "void") else len(split_args(args))
    defines = set(re.findall(r"#define\s+(YK\w+)", headers))
    types = set(re.findall(r"\}\s*(yk\w+_t)\s*;", headers)) | set(re.findall(r"typedef struct \w+ (yk\w+_t)\s*;", headers))
    problems, used_fn, used_def, used_types = [], set(), set(), set()
    go_dir = os.path.join(ROOT, "integration")
    for name in sorted(os.listdir(go_dir)):
        if not name.endswith(".go"):
            continue
        go = strip_comments(open(os.path.join(go_dir, name)).read())
        if "..." in re.sub(r'"[^"\n]*"', '""', go):
            problems.append(f"{name}: contains an elision ('...')")
        for m in re.finditer(r"\bC\.(\w+)", go):
            sym = m.group(1)
            after = go[m.end():m.end() + 1]
            if sym in LIBC:
                continue
            if sym.startswith("YK"):
                used_def.add(sym)
                if sym not in defines:
                    problems.append(f"{name}: C.{sym} is not #defined in include/")
            elif sym.endswith("_t"):
                used_types.add(sym)
                if sym not in types:
                    problems.append(f"{name}: type C.{sym} is not declared in include/")
            elif after == "(":
                used_fn.add(sym)
                if sym not in functions:
                    problems.append(f"{name}: C.{sym}() is not declared in include/")
                    continue
                got = len(split_args(call_args(go, m.end())))
                if got != functions[sym]:
                    problems.append(f"{name}: C.{sym}() called with {got} arguments, the header declares {functions[sym]}")
            else:
                problems.append(f"{name}: unrecognised C reference C.{sym}")
    print(f"checked {len(used_fn)} functions, {len(used_def)} constants, {len(used_types)} types of integration/*.go against include/*.h")
    for p in problems:
        print("PROBLEM:", p)
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
