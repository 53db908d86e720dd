#!/usr/bin/env python3
"""Re-flow the prose of a markdown file at a column limit (default 118): consecutive prose lines are joined into paragraphs /
bullet items and wrapped again with a hanging indent; headings, tables, code fences and blank lines are left alone.
usage: tools/wrap_md.py in.md out.md [cols]"""
import re
import sys
import textwrap

BULLET = re.compile(r"^(\s*)([*\-+] |\d+\. )")


def wrap(text, cols=118):
    out, fence, para = [], False, []

    def flush():
        if not para:
            return
        first = para[0]
        m = BULLET.match(first)
        if m:
            lead, bullet = m.group(1), m.group(2)
        else:
            lead, bullet = re.match(r"^(\s*)", first).group(1), ""
        body = " ".join([first[len(lead) + len(bullet):].strip()] + [p.strip() for p in para[1:]])
        out.extend(textwrap.wrap(body, cols, initial_indent=lead + bullet, subsequent_indent=lead + " " * len(bullet),
                                 break_long_words=False, break_on_hyphens=False))
        del para[:]
    for line in text.split("\n"):
        s = line.strip()
        if s.startswith("```"):
            flush()
            fence = not fence
            out.append(line)
        elif fence or not s or s.startswith("#") or (s.startswith("|") and s.endswith("|")) or s.startswith(">"):
            flush()
            out.append(line)
        elif BULLET.match(line):
            flush()
            para.append(line)
        else:
            para.append(line)
    flush()
    return "\n".join(out)


if __name__ == "__main__":
    cols = int(sys.argv[3]) if len(sys.argv) > 3 else 118
    open(sys.argv[2], "w").write(wrap(open(sys.argv[1]).read(), cols))
