"""Per kernel: bytes it moves per launch by the PMC counters (profiles/<tag>_f_pmc_hbm_traffic.txt: FETCH_SIZE x2 + WRITE_SIZE) over its
average duration when it runs alone (profiles/<tag>_e_kernel_stats_single_stream.txt), beside what a plain torch elementwise kernel
moves COLD at that size (tools/bw_probe_cold.py on the same pool: copy / add of 25..200 MB tensors, 640 MB of other traffic between
launches).  usage: python tools/insitu_rates.py round3 > profiles/round3_insitu_rates.txt"""
import re, sys
tag = sys.argv[1] if len(sys.argv) > 1 else 'round3'
COLD = [(50, 2.27), (100, 3.4), (150, 3.4), (200, 3.9), (300, 4.2), (400, 4.5), (600, 4.66)]     # MB moved -> TB/s (bw_probe_cold.py)
def plain(mb):
    if mb <= COLD[0][0]: return COLD[0][1]
    for (a, ra), (b, rb) in zip(COLD, COLD[1:]):
        if mb <= b: return ra + (rb - ra) * (mb - a) / (b - a)
    return COLD[-1][1]
dur = {}
for l in open('profiles/%s_e_kernel_stats_single_stream.txt' % tag):
    m = re.match(r'\s*([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(.*)', l)
    if m: dur[m.group(5).strip()] = (float(m.group(2)), float(m.group(3)))
print('# %s: bytes per launch (PMC) / duration alone (single-stream pass) vs the cold rate of a plain elementwise kernel of that size' % tag)
print('# %-62s %6s %8s %8s %7s %9s %6s' % ('kernel', 'calls', 'avg us', 'MB/call', 'TB/s', 'plain TB/s', 'ratio'))
tot_t = tot_p = 0.0
for l in open('profiles/%s_f_pmc_hbm_traffic.txt' % tag):
    m = re.match(r'\s*([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(.*)', l)
    if not m: continue
    name = m.group(5).strip()
    base = name.split('(')[0].strip()
    key = [k for k in dur if k.split('(')[0].strip() == base or (len(base) >= 58 and k.startswith(base[:58]))]
    if not key or 'fillBuffer' in name: continue
    calls, avg = dur[key[0]]
    if abs(calls - float(m.group(4))) > 0.25 * calls: continue       # (launch counts of the two passes differ: not the same set of launches)
    mb = (float(m.group(2)) + float(m.group(3))) / float(m.group(4))
    if mb < 20: continue                        # latency-bound launches: no bandwidth statement
    rate = mb / avg                             # MB/us = TB/s
    print('  %-62s %6.1f %8.1f %8.1f %7.2f %9.2f %6.2f' % (name[:62], calls, avg, mb, rate, plain(mb), rate / plain(mb)))
    tot_t += calls * avg; tot_p += calls * mb / plain(mb)
print('# listed launches: %.2f ms alone per step; the same bytes in plain cold kernels of the same sizes: %.2f ms' % (tot_t / 1e3, tot_p / 1e3))
