#!/usr/bin/env python
"""Re-wrap the prose of a markdown file at 120 columns (tables, code fences, headings and short lines are left alone; list items keep
their hanging indent).  Table rows whose cells hold paragraphs are turned into list entries: `| a | b | c |` -> `* **a** — b — c`.
    python tools/wrap_md.py DESIGN.md [--tables-to-lists]"""
import re
import sys
import textwrap

WIDTH = 120


def wrap_line(line):
    m = re.match(r'^(\s*(?:[*\-+]|\d+\.)\s+|\s*>\s?|\s+)?(.*)$', line)
    lead, body = m.group(1) or '', m.group(2)
    hang = ' ' * len(lead) if lead.strip() else lead
    return textwrap.fill(body, WIDTH, initial_indent=lead, subsequent_indent=hang, break_long_words=False, break_on_hyphens=False)


def main():
    path = sys.argv[1]
    to_lists = '--tables-to-lists' in sys.argv
    out, fence = [], False
    for line in open(path).read().split('\n'):
        if line.strip().startswith('```'):
            fence = not fence
            out.append(line)
            continue
        if fence or len(line) <= WIDTH + 10 or line.startswith('#'):
            out.append(line)
            continue
        if line.lstrip().startswith('|'):
            cells = [c.strip() for c in line.strip().strip('|').split('|')]
            if to_lists and not all(re.fullmatch(r':?-+:?', c) for c in cells):
                head = '* **%s**' % cells[0] if cells[0] else '*'
                out.append(wrap_line(head + ' — ' + ' — '.join(c for c in cells[1:] if c)))
            elif not to_lists:
                out.append(line)
            continue
        out.append(wrap_line(line))
    open(path, 'w').write('\n'.join(out))


if __name__ == '__main__':
    main()
