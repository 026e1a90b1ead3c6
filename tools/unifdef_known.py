"""Dev tool: resolve the preprocessor conditionals of a source for a set of KNOWN macros and substitute their values, leaving
every other conditional untouched (a small `unifdef`, which this image does not have).  Used in round 6 to take rejected A/B
variants out of the product sources; the removed text is kept as reverse diffs under tools/patches/.

    python tools/unifdef_known.py FILE NAME=VALUE ... NAME- ...      (NAME- : known to be UNDEFINED)   -> rewrites FILE in place
"""
import re
import sys


def _eval(expr, known):
    e = re.sub(r"/\*.*?\*/", " ", expr)
    e = re.sub(r"//.*$", "", e).strip()
    e = re.sub(r"defined\s*\(\s*(\w+)\s*\)|defined\s+(\w+)",
               lambda m: (("1" if known[m.group(1) or m.group(2)] is not None else "0") if (m.group(1) or m.group(2)) in known
                          else m.group(0)), e)
    if "defined" in e:
        return None
    ids = set(re.findall(r"\b[A-Za-z_]\w*\b", e))
    if not ids <= set(known):
        return None
    for n in ids:
        e = re.sub(r"\b%s\b" % n, "0" if known[n] is None else str(known[n]), e)
    e = e.replace("&&", " and ").replace("||", " or ").replace("!", " not ").replace(" not =", "!=")
    try:
        return bool(eval(e, {"__builtins__": {}}, {}))
    except Exception:
        return None


def process(text, known):
    out, stack = [], []            # frame: [resolved?, any branch taken, this branch active, emitting before the frame]
    emitting = True
    for line in text.splitlines(True):
        m = re.match(r"\s*#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)", line)
        if not m:
            if emitting:
                out.append(line)
            continue
        d, rest = m.group(1), m.group(2)
        if d in ("if", "ifdef", "ifndef"):
            name = rest.split()[0] if rest.split() else ""
            val = _eval(rest, known) if d == "if" else ((known[name] is not None) == (d == "ifdef") if name in known else None)
            resolved = val is not None and emitting
            stack.append([resolved, bool(val), bool(val), emitting])
            if resolved:
                emitting = bool(val)
            elif emitting:
                out.append(line)
        elif d == "elif":
            f = stack[-1]
            if f[0]:
                val = _eval(rest, known)
                if val is None:
                    raise SystemExit(f"#elif with unknown macros after a resolved #if: {line.strip()}")
                f[2] = (not f[1]) and val
                f[1] = f[1] or val
                emitting = f[3] and f[2]
            elif f[3]:
                out.append(line)
        elif d == "else":
            f = stack[-1]
            if f[0]:
                f[2] = not f[1]
                emitting = f[3] and f[2]
            elif f[3]:
                out.append(line)
        else:
            f = stack.pop()
            if not f[0] and f[3]:
                out.append(line)
            emitting = f[3]
    res = "".join(out)
    for n, v in known.items():
        if v is not None:
            res = re.sub(r"\b%s\b" % n, str(v), res)
    return res


if __name__ == "__main__":
    path, known = sys.argv[1], {}
    for a in sys.argv[2:]:
        if a.endswith("-"):
            known[a[:-1]] = None
        else:
            k, v = a.split("=", 1)
            known[k] = v
    src = open(path).read()
    new = process(src, known)
    open(path, "w").write(new)
    print(f"{path}: {src.count(chr(10))} -> {new.count(chr(10))} lines")
