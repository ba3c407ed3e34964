"""tools/pmc_traffic.py outdir run_dir -- HBM-side traffic and MFMA utilisation per kernel of one bench step from the
rocprofv3 --pmc passes of tools/gpu/r02_p.sh (separate passes, --kernel-trace only, as MI355X_MICROARCH.md prescribes):
FETCH_SIZE / WRITE_SIZE are reported in KB; FETCH_SIZE is doubled (gfx950: 128-byte requests tallied at 64 B for wide
coalesced reads), WRITE_SIZE taken as reported.  Per C-ABI entry: bytes per launch (mean over the step's launches)."""
import collections, csv, glob, json, os, sys

out_dir, run = sys.argv[1], sys.argv[2]
ENTRY = [('so3_group_lists_kernel<true', 'eap_so3_inter_group_inv_f32'), ('so3_group_lists_kernel<false, 2', 'eap_so3_inter_group_fwd_t_f32'),
         ('gemm_dma_f32_kernel', 'eap_gemm_dma_f32'), ('zpconv_mfma_kernel', 'eap_inter_zpconv_fwd_ws_f32 (matrix kernel)'), ('zpconv_index_check_kernel', 'eap_inter_zpconv_*_ws_f32 (index check)'),
         ('zpconv_bwd_t_kernel', 'eap_inter_zpconv_bwd_ws_f32 (products)'), ('zpconv_bwd_sum_kernel', 'eap_inter_zpconv_bwd_ws_f32 (sums)'),
         ('bn_act_bwd_apply', 'eap_bn_act_bwd_apply_f32'), ('bn_act_fwd', 'eap_bn_act_fwd_f32')]


def entry_of(name):
    for pat, e in ENTRY:
        if pat in name:
            return e
    return None


def collect(sub):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(dict)
    for f in glob.glob(os.path.join(run, sub, '*counter_collection.csv')):
        for r in csv.DictReader(open(f)):
            e = entry_of(r['Kernel_Name'])
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
per, dur = collect('pmc_mfma')
for e, c in per.items():
    gui = per_dispatch_sum(c['GRBM_GUI_ACTIVE']); mf = per_dispatch_sum(c['SQ_VALU_MFMA_BUSY_CYCLES'])
    cyc = sum(gui.values()) / 8.0                            # GRBM_GUI_ACTIVE is summed over the 8 XCDs
    res.setdefault(e, {})['mfma_util'] = sum(mf.values()) / (cyc * 1024) if cyc else None
    res[e]['shader_clock_ghz'] = cyc / max(sum(dur[e].values()), 1)
    res[e]['avg_launch_ms_under_pmc'] = sum(dur[e].values()) / len(dur[e]) / 1e6
per, _ = collect('pmc_waves')
for e, c in per.items():
    tot = sum(v for _, v in c['SQ_WAVE_CYCLES'])
    for n in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY'):
        res.setdefault(e, {})['frac_' + n] = sum(v for _, v in c[n]) / tot if tot else None
per_launch = {}
for e, r in res.items():
    if 'FETCH_SIZE_KB_per_launch' in r:
        per_launch[e] = {'fetch': 2.0 * 1024 * r['FETCH_SIZE_KB_per_launch'], 'write': 1024 * r.get('WRITE_SIZE_KB_per_launch', 0.0),
                         'l2_hit': r.get('l2_hit')}
doc = {'how': __doc__.strip(), 'per_launch_bytes': per_launch, 'counters': res}
json.dump(doc, open(os.path.join(out_dir, 'r02_pmc_traffic.json'), 'w'), indent=1)
print(json.dumps(doc['counters'], indent=1))
