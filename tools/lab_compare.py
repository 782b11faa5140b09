#!/usr/bin/env python3
"""Compares outputs of tools/conv_lab2 builds run on one box: lab_compare.py base.txt[,base2.txt] new.txt[,new2.txt] ...
Per shape the best time of each variant (plain / fused reduction / statistics), relative to the first argument."""
import re
import sys


def parse(f):
    out = {}
    for l in open(f):
        m = re.search(r'plain\s+([\d.]+) us.*bred\s+([\d.]+) us.*?\]\s+(\S+)\s+(\d+)->(\d+).*pipe\s+([\d.]+) us.*mismatches (\d+).*bred sums (\S+) (\S+)', l)
        if m:
            out[(m.group(3), m.group(4), m.group(5))] = (float(m.group(1)), float(m.group(2)), float(m.group(6)), m.group(7), m.group(8), m.group(9))
    return out


def best(files):
    runs = [parse(f) for f in files.split(',')]
    return {k: tuple(min(r[k][i] for r in runs) for i in range(3)) + runs[0][k][3:] for k in runs[0]}


base = best(sys.argv[1])
for arg in sys.argv[2:]:
    new = best(arg)
    print('==', arg)
    tot_o, tot_n = [0, 0, 0], [0, 0, 0]
    for k in base:
        o, n = base[k], new[k]
        for i in range(3):
            tot_o[i] += o[i]
            tot_n[i] += n[i]
        print('%-9s %5s->%-5s plain %6.1f->%6.1f (%+5.1f%%) bred %6.1f->%6.1f (%+5.1f%%) stats %6.1f->%6.1f (%+5.1f%%) mismatches %s %s' % (
            k[0], k[1], k[2], o[0], n[0], 100 * (n[0] / o[0] - 1), o[1], n[1], 100 * (n[1] / o[1] - 1), o[2], n[2], 100 * (n[2] / o[2] - 1),
            n[3], 'sums same' if n[4:] == o[4:] else 'SUMS DIFFER'))
    print('total: plain %+.1f%% bred %+.1f%% stats %+.1f%%' % tuple(100 * (tot_n[i] / tot_o[i] - 1) for i in range(3)))
