"""Re-flow the long prose lines of a Markdown file to a fixed width (tables, code fences and headings are left alone):
    python tools/md_reflow.py DESIGN.md [width=150]
List items keep their marker and get a hanging indent; nothing else about the text changes."""
import re
import sys
import textwrap


def reflow(text, width):
    out, fence = [], False
    for line in text.split("\n"):
        if line.lstrip().startswith("```"):
            fence = not fence
        if fence or len(line) <= width or line.lstrip().startswith(("|", "#", "```")):
            out.append(line)
            continue
        m = re.match(r"^(\s*)((?:[-*+]|\d+\.)\s+)?", line)
        lead, marker = m.group(1), m.group(2) or ""
        body = line[len(lead) + len(marker):]
        out.extend(textwrap.wrap(body, width=width, initial_indent=lead + marker, subsequent_indent=lead + " " * len(marker), break_long_words=False, break_on_hyphens=False))
    return "\n".join(out)


if __name__ == "__main__":
    path = sys.argv[1]
    width = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    src = open(path).read()
    open(path, "w").write(reflow(src, width))
