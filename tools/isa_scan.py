#!/usr/bin/env python3
"""Scan a hipcc -S listing: for each kernel whose name contains one of the given substrings, print the loop labels, branches,
barriers, spill traffic (scratch_/v_writelane/v_readlane), LDS-DMA and vmcnt waits in program order, plus instruction counts.
Usage: hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only file.hip -o file.s ; tools/isa_scan.py file.s name [name...]"""
import re
import sys
from collections import Counter


def main():
    text = open(sys.argv[1]).read()
    for name in sys.argv[2:]:
        for m in re.finditer(r'^(_Z\S*' + re.escape(name) + r'\S*):\s*.*?\n(.*?)^\.Lfunc_end', text, re.S | re.M):
            lines = m.group(2).split('\n')
            ops = Counter(l.split()[0] for l in lines if l.startswith('\t') and not l.strip().startswith(('.', ';')))
            print('==', m.group(1)[:90], 'instructions', sum(ops.values()))
            pk = sum(v for k, v in ops.items() if k.startswith('v_pk_'))
            valu = sum(v for k, v in ops.items() if k.startswith('v_'))
            print('   v_pk_*', pk, ' v_*', valu, ' ds_*', sum(v for k, v in ops.items() if k.startswith('ds_')),
                  ' s_barrier', ops.get('s_barrier', 0), ' scratch', sum(v for k, v in ops.items() if k.startswith('scratch_')))
            for i, l in enumerate(lines):
                t = l.strip()
                if any(x in t for x in ('scratch_', 'v_writelane', 'v_readlane', 's_barrier', 's_cbranch', 'global_load_lds', 's_waitcnt vmcnt')) or t.startswith('.LBB'):
                    print('  ', i, t)


if __name__ == '__main__':
    main()
