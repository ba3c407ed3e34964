"""tools/pmc_traffic.py out_dir run_dir tag -- fabric-side traffic and matrix-pipe utilisation PER HIP KERNEL of one bench
step, from the rocprofv3 --pmc passes of tools/gpu/profile_round.sh (separate passes, --kernel-trace only, as
MI355X_MICROARCH.md prescribes).  FETCH_SIZE / WRITE_SIZE are reported in KB; FETCH_SIZE is doubled (gfx950: 128-byte
requests tallied at 64 B for wide coalesced reads), WRITE_SIZE taken as reported.  Kernels are keyed by the start of
their demangled name (namespace prefix stripped, template arguments kept) -- the key bench.py looks its kernels up by.
Written to out_dir/<tag>_pmc_traffic.json (copy into profiles/ to have bench.py report `traffic`)."""
import collections, csv, glob, json, os, re, sys

out_dir, run, tag = sys.argv[1], sys.argv[2], sys.argv[3]
WANT = ('kc_gemm_kernel', 'dense_', 'so3_group_lists', 'gemm_bf16x3_kernel', 'gemm_f16x2_kernel', 'gemm_dma_f32_kernel', 'gemm_f32_kernel', 'zpconv_', 'zp_hot', 'bn_act_', 'so3_inter_group', 'chamfer', 'anchor_attn')


def key_of(name):
    """'void (anonymous namespace)::gemm_dma_f32_kernel<2, 2, 4, 4, false, false>(args)' -> 'gemm_dma_f32_kernel<2, 2, 4, 4, false, false>'"""
    n = re.sub(r'^void\s+', '', name)
    n = n.replace('(anonymous namespace)::', '')
    depth, out = 0, ''
    for ch in n:                      # cut at the argument list: the first '(' outside template brackets
        if ch == '<':
            depth += 1
        elif ch == '>':
            depth -= 1
        elif ch == '(' and depth == 0:
            break
        out += ch
    if out.startswith('gemm_dma_f32_kernel') or out.startswith('kc_gemm_kernel'):
        out = re.sub(r',\s*0>$', '>', out)      # trailing default template argument of the GEMM (ablation switch)
    m = re.match(r'(gemm_bf16x3_kernel|gemm_f16x2_kernel)<(\d+), (\d+), \d+, (\d+)(?:, (?:true|false))?>$', out)
    if m:                                       # <MI, WN, ablation switch, B layout, pre-split weights> -> the name eap_last_kernel() reports
        out = '%s<%s, %s%s>' % (m.group(1), m.group(2), m.group(3), ('', ', nn', ', gather')[int(m.group(4))])
    return out if any(w in out for w in WANT) else None


def collect(sub):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(dict)
    for f in glob.glob(os.path.join(run, sub, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            e = key_of(r['Kernel_Name'])
            if e:
                per[e][r['Counter_Name']].append((r['Dispatch_Id'], float(r['Counter_Value'])))
                dur[e][r['Dispatch_Id']] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    return per, dur


def per_dispatch_sum(vals):
    d = collections.defaultdict(float)
    for disp, v in vals:
        d[disp] += v
    return d


res = {}
for sub, ctr in (('pmc_FETCH_SIZE', 'FETCH_SIZE'), ('pmc_WRITE_SIZE', 'WRITE_SIZE')):
    per, _ = collect(sub)
    for e, c in per.items():
        d = per_dispatch_sum(c[ctr])
        res.setdefault(e, {})[ctr + '_KB_per_launch'] = sum(d.values()) / len(d)
        res[e]['launches_per_step'] = len(d) / 2.0          # the run is 1 warm-up + 1 timed step
per, _ = collect('pmc_TCC_HIT_sum_TCC_MISS_sum')
for e, c in per.items():
    h, m = sum(v for _, v in c['TCC_HIT_sum']), sum(v for _, v in c['TCC_MISS_sum'])
    res.setdefault(e, {})['l2_hit'] = h / max(h + m, 1.0)
per, dur = collect('pmc_SQ_VALU_MFMA_BUSY_CYCLES_SQ_BUSY_CU_CYCLES_GRBM_GUI_ACTIVE_SQ_WAVES')
for e, c in per.items():
    gui = per_dispatch_sum(c['GRBM_GUI_ACTIVE']); mf = per_dispatch_sum(c['SQ_VALU_MFMA_BUSY_CYCLES'])
    cyc = sum(gui.values()) / 8.0                            # GRBM_GUI_ACTIVE is summed over the 8 XCDs
    res.setdefault(e, {})['mfma_util'] = sum(mf.values()) / (cyc * 1024) if cyc else None
    res[e]['shader_clock_ghz'] = cyc / max(sum(dur[e].values()), 1)
    res[e]['avg_launch_ms_under_pmc'] = sum(dur[e].values()) / len(dur[e]) / 1e6
# wave-cycle breakdown (optional passes pmc_breakdown_*): per kernel, counter sums per launch and, where SQ_WAVE_CYCLES is in the
# same pass, as fractions of it (WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES, MI355X_MICROARCH.md)
breakdown = {}
for d in sorted(glob.glob(os.path.join(run, 'pmc_breakdown_*'))):
    per, _ = collect(os.path.basename(d))
    for e, c in per.items():
        tot = {name: sum(v for _, v in vals) / max(len(per_dispatch_sum(vals)), 1) for name, vals in c.items()}
        b = breakdown.setdefault(e, {})
        b.update({k: v for k, v in tot.items()})
        if 'SQ_WAVE_CYCLES' in tot and tot['SQ_WAVE_CYCLES'] > 0:
            b.update({k + '_frac_of_wave_cycles': v / tot['SQ_WAVE_CYCLES'] for k, v in tot.items() if k != 'SQ_WAVE_CYCLES'})
per_kernel = {}
for e, r in res.items():
    if 'FETCH_SIZE_KB_per_launch' in r:
        per_kernel[e] = {'fetch': 2.0 * 1024 * r['FETCH_SIZE_KB_per_launch'], 'write': 1024 * r.get('WRITE_SIZE_KB_per_launch', 0.0),
                         'l2_hit': r.get('l2_hit'), 'mfma_util': r.get('mfma_util'), 'shader_clock_ghz': r.get('shader_clock_ghz'),
                         'launches_per_step': r.get('launches_per_step')}
import datetime
doc = {'how': __doc__.strip(), 'collected': datetime.date.today().isoformat(), 'per_kernel': per_kernel, 'counters': res, 'wave_cycle_breakdown': breakdown}
json.dump(doc, open(os.path.join(out_dir, f'{tag}_pmc_traffic.json'), 'w'), indent=1)
print(json.dumps(per_kernel, indent=1))
