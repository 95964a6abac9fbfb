#!/usr/bin/env python3
"""Fingerprint of the kernel sources: sha256 over every .hpp / .hip / .inc of this directory and include/uvs_rmckf.h with comments and blank
space removed, first 12 hex digits.  The Makefile compiles it into uvs_version(); tools/make_traffic_json.py records it next to the counter
figures of profiles/traffic_latest.json; tests/test_host_logic.py and bench.py compare the three, so that PMC counts cannot outlive a kernel
change (a comment edit does not move it)."""
import glob
import hashlib
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))


def stripped(text):
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    text = re.sub(r'//[^\n]*', '', text)
    return '\n'.join(' '.join(line.split()) for line in text.splitlines() if line.strip())


def source_hash():
    files = sorted(glob.glob(os.path.join(HERE, '*.hpp')) + glob.glob(os.path.join(HERE, '*.hip')) + glob.glob(os.path.join(HERE, '*.inc')))
    files.append(os.path.join(HERE, '..', '..', 'include', 'uvs_rmckf.h'))
    h = hashlib.sha256()
    for path in files:
        h.update(os.path.basename(path).encode() + b'\0')
        h.update(stripped(open(path, encoding='utf-8').read()).encode() + b'\0')
    return h.hexdigest()[:12]


if __name__ == '__main__':
    print(source_hash())
