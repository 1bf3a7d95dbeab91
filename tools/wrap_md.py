#!/usr/bin/env python3
"""Re-wrap the prose of a Markdown file at a column limit (VERDICT r05 hygiene: DESIGN.md at ~130 columns).
Paragraphs and list items are re-flowed with a hanging indent; headings, tables, fenced code and blank lines are left alone.
    python tools/wrap_md.py DESIGN.md [130]"""
import re
import sys
import textwrap

path = sys.argv[1]
width = int(sys.argv[2]) if len(sys.argv) > 2 else 130
lines = open(path).read().split("\n")
out, para, in_code = [], [], False


def flush():
    global para
    if not para:
        return
    first = para[0]
    m = re.match(r"^(\s*)([*+-]|\d+\.)\s+", first)
    if m:
        lead = first[: m.end()]
        hang = " " * len(lead)
        text = " ".join([first[m.end():].strip()] + [x.strip() for x in para[1:]])
    else:
        lead = re.match(r"^\s*", first).group(0)
        hang = lead
        text = " ".join(x.strip() for x in para)
    out.extend(textwrap.wrap(text, width=width, initial_indent=lead, subsequent_indent=hang, break_long_words=False, break_on_hyphens=False))
    para = []


for ln in lines:
    if ln.strip().startswith("```"):
        flush()
        in_code = not in_code
        out.append(ln)
        continue
    if in_code or ln.startswith("|") or ln.startswith("#") or not ln.strip():
        flush()
        out.append(ln)
        continue
    if re.match(r"^\s*([*+-]|\d+\.)\s+", ln):      # a new list item ends the previous block
        flush()
    para.append(ln)
flush()
open(path, "w").write("\n".join(out))
