"""Re-flow the prose of a Markdown file to a fixed width (tables, code fences, headings and blank lines are left alone):
    python tools/md_reflow.py DESIGN.md [width=150]
Consecutive lines of a paragraph or of a list item are joined first, then wrapped; list items keep their marker and get a hanging indent.
The words of the file and their order do not change."""
import re
import sys
import textwrap

ITEM = re.compile(r"^(\s*)((?:[-*+]|\d+\.)\s+)")


def reflow(text, width):
    out, fence, cur = [], False, None          # cur = [lead, marker, words...] of the paragraph / item being collected

    def flush():
        nonlocal cur
        if cur is not None:
            lead, marker, body = cur
            out.extend(textwrap.wrap(" ".join(body), width=width, initial_indent=lead + marker, subsequent_indent=lead + " " * len(marker),
                                     break_long_words=False, break_on_hyphens=False))
            cur = None
    for line in text.split("\n"):
        st = line.lstrip()
        if st.startswith("```"):
            flush(); fence = not fence; out.append(line); continue
        if fence or not st or st.startswith(("|", "#")):
            flush(); out.append(line); continue
        m = ITEM.match(line)
        if m:
            flush(); cur = [m.group(1), m.group(2), [line[m.end():].strip()]]
        elif cur is not None:
            cur[2].append(st)
        else:
            cur = [line[:len(line) - len(st)], "", [st]]
    flush()
    return "\n".join(out)


if __name__ == "__main__":
    path = sys.argv[1]
    width = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    src = open(path).read()
    open(path, "w").write(reflow(src, width))
