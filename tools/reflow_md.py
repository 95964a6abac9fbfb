#!/usr/bin/env python3
"""Keep the Markdown documents readable in a diff and in a terminal: no line over WIDTH characters.

  * a paragraph or list item that runs long is re-wrapped (continuation lines of a list item are indented under its text);
  * a table with a row that runs long cannot be wrapped as a table: it becomes a list -- one item per row, the first cell in bold, the remaining
    cells (prefixed by their column's header where the table has more than two columns) as wrapped text under it;
  * fenced code blocks and tables whose rows all fit are left alone.

usage: tools/reflow_md.py [--check] file.md ...     (--check: report the longest line of each file and exit 1 if one exceeds WIDTH)"""
import re
import sys
import textwrap

WIDTH = 160


def cells(row):
    row = row.strip()
    if row.startswith('|'):
        row = row[1:]
    if row.endswith('|'):
        row = row[:-1]
    return [c.strip() for c in re.split(r'(?<!\\)\|', row)]


def wrap(text, first, rest):
    return textwrap.wrap(text, WIDTH, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False) or [first.rstrip()]


def table_to_list(rows):
    header = cells(rows[0])
    out = []
    for row in rows[2:]:
        c = cells(row)
        head = c[0] if c[0].startswith('**') or not c[0] else f'**{c[0]}**'
        body = []
        for name, val in zip(header[1:], c[1:]):
            if not val:
                continue
            body.append(f'{name}: {val}' if len(header) > 2 else val)
        text = head + (' — ' + '; '.join(body) if body else '')
        out += wrap(text, '- ', '  ')
    return out


def reflow(lines):
    out, i, n = [], 0, len(lines)
    while i < n:
        line = lines[i]
        if line.lstrip().startswith('```'):                      # fenced block: verbatim
            j = i + 1
            while j < n and not lines[j].lstrip().startswith('```'):
                j += 1
            out += lines[i:j + 1]
            i = j + 1
            continue
        if line.lstrip().startswith('|') and i + 1 < n and re.match(r'^\s*\|?\s*:?-{2,}', lines[i + 1]):
            j = i
            while j < n and lines[j].lstrip().startswith('|'):
                j += 1
            rows = lines[i:j]
            if max(len(r) for r in rows) > WIDTH:
                lead = ' | '.join(cells(rows[0]))
                out += wrap(f'({lead}:)', '', '') + [''] + table_to_list(rows) + ['']
            else:
                out += rows
            i = j
            continue
        if line.startswith('#'):
            if len(line) <= WIDTH:
                out.append(line)
            else:                                                # an over-long heading: keep its first part as the heading, the rest as text
                head, _, rest = line.partition(' — ')
                out.append(head)
                if rest:
                    out += [''] + wrap(rest, '', '')
            i += 1
            continue
        if not line.strip():
            out.append(line)
            i += 1
            continue
        # a paragraph or a list item with its continuation lines: re-wrapped as a whole when one of its lines runs long
        m = re.match(r'^(\s*)([-*+]|\d+\.)\s+', line)
        j = i + 1
        while j < n and lines[j].strip() and not lines[j].lstrip().startswith(('|', '#', '```')) and not re.match(r'^\s*([-*+]|\d+\.)\s+', lines[j]):
            j += 1
        block = lines[i:j]
        if max(len(x) for x in block) <= WIDTH:
            out += block
        elif m:
            text = ' '.join([line[m.end():].strip()] + [x.strip() for x in block[1:]])
            out += wrap(text, m.group(0), ' ' * len(m.group(0)))
        else:
            lead = re.match(r'^\s*', line).group(0)
            out += wrap(' '.join(x.strip() for x in block), lead, lead)
        i = j
    return out


def main():
    check = '--check' in sys.argv
    bad = False
    for path in [a for a in sys.argv[1:] if not a.startswith('--')]:
        lines = open(path, encoding='utf-8').read().split('\n')
        if check:
            longest = max(len(x) for x in lines)
            print(f'{path}: longest line {longest}')
            bad |= longest > WIDTH
            continue
        new = reflow(lines)
        open(path, 'w', encoding='utf-8').write('\n'.join(new))
        print(f'{path}: {len(lines)} -> {len(new)} lines, longest {max(len(x) for x in new)}')
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
