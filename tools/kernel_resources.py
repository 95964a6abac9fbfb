#!/usr/bin/env python3
"""Register / scratch / LDS budget of every kernel in libuvs_rmckf.so, read from the gfx950 code objects embedded in the library
(clang offload bundles in .hip_fatbin -> ELF notes via llvm-readelf).  usage: tools/kernel_resources.py [--lib PATH] [filter]"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READELF = '/opt/rocm/lib/llvm/bin/llvm-readelf'
FILT = 'c++filt'
MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'


def code_objects(lib):
    data = open(lib, 'rb').read()
    for m in re.finditer(MAGIC, data):
        base = m.start()
        (n,) = struct.unpack_from('<Q', data, base + len(MAGIC))
        pos = base + len(MAGIC) + 8
        for _ in range(n):
            off, size, tsz = struct.unpack_from('<QQQ', data, pos)
            triple = data[pos + 24: pos + 24 + tsz].decode()
            pos += 24 + tsz
            if 'gfx950' in triple and size:
                yield data[base + off: base + off + size]


def kernels(lib=None):
    """{demangled kernel name: dict(vgpr, agpr, sgpr, scratch, lds, wg)}"""
    lib = lib or os.path.join(ROOT, 'uncalibrated-visual-servoing_amd', 'libuvs_rmckf.so')
    out = {}
    for blob in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix='.co') as fh:
            fh.write(blob)
            fh.flush()
            notes = subprocess.run([READELF, '--notes', fh.name], capture_output=True, text=True, check=True).stdout
        for block in notes.split('- .agpr_count:')[1:]:
            get = lambda key: int(re.search(r'\.' + key + r':\s+(\d+)', block).group(1))       # noqa: E731
            agpr = int(block.split('\n', 1)[0].strip())
            name = re.search(r'\.name:\s+(\S+)', block).group(1)
            out[name] = dict(vgpr=get('vgpr_count'), agpr=agpr, sgpr=get('sgpr_count'), scratch=get('private_segment_fixed_size'),
                             lds=get('group_segment_fixed_size'), wg=get('max_flat_workgroup_size'))
    names = list(out)
    dem = subprocess.run([FILT], input='\n'.join(names), capture_output=True, text=True, check=True).stdout.split('\n')
    return {re.sub(r'^void uvs::|\(uvs::\w+\)$', '', d): out[n] for n, d in zip(names, dem)}


if __name__ == '__main__':
    args = [a for a in sys.argv[1:]]
    lib = None
    if '--lib' in args:
        lib = args[args.index('--lib') + 1]
        del args[args.index('--lib'): args.index('--lib') + 2]
    pat = args[0] if args else ''
    print(f'{"kernel":100s} {"vgpr":>5s} {"agpr":>5s} {"sgpr":>5s} {"scratch B":>9s} {"LDS B":>7s}')
    for name, r in sorted(kernels(lib).items()):
        if pat in name:
            print(f'{name[:100]:100s} {r["vgpr"]:5d} {r["agpr"]:5d} {r["sgpr"]:5d} {r["scratch"]:9d} {r["lds"]:7d}')
