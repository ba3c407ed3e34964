"""tools/pmc_summary.py dir... -- per kernel (launches >= 1 ms) of rocprofv3 --pmc CSV outputs: the counters summed
over the kernel's dispatches, MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs),
shader clock = GRBM_GUI_ACTIVE / 8 / duration, and the wave-state split."""
import collections, csv, glob, json, sys
out = {}
for d in sys.argv[1:]:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        dur = collections.defaultdict(dict)
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][-70:]
            per[k][r['Counter_Name']] += float(r['Counter_Value'])
            dur[k][r['Dispatch_Id']] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        for k, v in per.items():
            ns = sum(dur[k].values())
            if ns < 1e6 * len(dur[k]):
                continue
            e = out.setdefault(k, {'launches': len(dur[k]), 'avg_ms': ns / len(dur[k]) / 1e6})
            e.update({n: x for n, x in v.items()})
for k, e in out.items():
    if 'GRBM_GUI_ACTIVE' in e and e['GRBM_GUI_ACTIVE'] > 0:
        cyc = e['GRBM_GUI_ACTIVE'] / 8.0
        e['shader_clock_ghz'] = cyc / (e['avg_ms'] * e['launches'] * 1e6)
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in e:
            e['mfma_util'] = e['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024)
    if e.get('SQ_WAVE_CYCLES', 0) > 0:
        for n in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VMEM', 'SQ_ACTIVE_INST_VALU'):
            if n in e:
                e['frac_' + n] = e[n] / e['SQ_WAVE_CYCLES']
print(json.dumps(out, indent=1))
