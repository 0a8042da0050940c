#!/usr/bin/env python3
"""Re-wrap the prose of a Markdown file to a column limit (default 120) without touching tables, code fences, headings
or the relative indentation of list items.   tools/wrap_md.py DESIGN.md [width]"""
import re
import sys
import textwrap

path = sys.argv[1]
width = int(sys.argv[2]) if len(sys.argv) > 2 else 120
lines = open(path, encoding="utf-8").read().split("\n")
out, para, fence = [], [], False
bullet = re.compile(r"^(\s*)([*+-]|\d+\.)\s+")


def flush():
    if not para:
        return
    first = para[0]
    m = bullet.match(first)
    if m:
        lead = first[: m.end()]
        hang = " " * len(lead)
        text = " ".join([first[m.end():].strip()] + [p.strip() for p in para[1:]])
    else:
        ind = re.match(r"^\s*", first).group(0)
        lead = hang = ind
        text = " ".join(p.strip() for p in para)
    out.extend(textwrap.wrap(text, width=width, initial_indent=lead, subsequent_indent=hang, break_long_words=False,
                             break_on_hyphens=False) or [lead.rstrip()])
    para.clear()


for ln in lines:
    st = ln.strip()
    if st.startswith("```"):
        flush()
        fence = not fence
        out.append(ln)
        continue
    if fence or st.startswith("|") or st.startswith("#") or st == "" or st.startswith("<"):
        flush()
        out.append(ln)
        continue
    if bullet.match(ln) and para:
        flush()
    elif para and not bullet.match(ln):
        # a continuation line: belongs to the running paragraph only if it is indented at least like its text
        pass
    para.append(ln)
flush()
open(path, "w", encoding="utf-8").write("\n".join(out))
print(sum(1 for l in out if len(l) > width and not l.strip().startswith("|")), "prose lines still over", width)
