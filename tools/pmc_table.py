"""Join rocprofv3 --pmc passes per dispatch (same program => same dispatch order) and print
per-(kernel, grid, lds) averages with derived ratios."""
import csv, glob, collections, sys, json
passes = sys.argv[1:]
rows = collections.OrderedDict()   # dispatch_id -> dict
for p in passes:
    f = glob.glob(f'gpurun_out/pmc_{p}/*/*_counter_collection.csv')[0]
    for r in csv.DictReader(open(f)):
        d = int(r['Dispatch_Id'])
        e = rows.setdefault(d, {'name': r['Kernel_Name'], 'grid': int(r['Grid_Size']), 'lds': int(r['LDS_Block_Size']),
                                'vgpr': int(r['VGPR_Count']), 'agpr': int(r['Accum_VGPR_Count'])})
        e[r['Counter_Name']] = float(r['Counter_Value'])
        e.setdefault('dur_' + p, float(r['End_Timestamp']) - float(r['Start_Timestamp']))
# label dispatches of the LAST step using the op list of a bench run (same plan)
ops = json.load(open(sys.argv[0].replace('pmc_table.py', '../gpurun_out/ops_g0_default.json'))) if False else None
agg = collections.OrderedDict()
for d, e in rows.items():
    if 'k_conv_mfma' not in e['name'] and 'chan_stats' not in e['name'] and 'bgemm' not in e['name']: continue
    short = e['name'].split('(')[0].replace('void dsx::', '').replace('dsx::', '')[:60]
    k = (short, e['grid'], e['lds'], e['vgpr'], e['agpr'])
    a = agg.setdefault(k, collections.Counter())
    a['n'] += 1
    for c, v in e.items():
        if isinstance(v, float): a[c] += v
def g(a, c): return a.get(c, 0.0) / a['n']
print(f"{'kernel':40s} {'grid':>7s} {'lds':>6s} {'v+a':>7s} n  dur_us | wave_cyc/wave busy% wait_any% wait_inst% act_valu% mfma_busy% | valu/wave mfma/wave lds/wave vmemrd/wave salu/wave trans/wave | ldsconf% | fetchMB writeMB L2hit%")
for k, a in sorted(agg.items(), key=lambda kv: -g(kv[1], 'dur_A') * kv[1]['n'])[:28]:
    waves = g(a, 'SQ_WAVES') or 1
    wc = g(a, 'SQ_WAVE_CYCLES')
    def pct(c): return 100.0 * g(a, c) / wc if wc else 0
    dur = g(a, 'dur_A') / 1e3
    # SQ_WAVE_CYCLES etc. are in quad-cycles summed over waves; MFMA_BUSY in cycles (per SIMD?) -> relate to dur*2.4GHz*1024 SIMDs
    mfma_busy = 100.0 * g(a, 'SQ_VALU_MFMA_BUSY_CYCLES') / (dur * 1e-6 * 2.4e9 * 1024) if dur else 0
    print(f"{k[0][:40]:40s} {k[1]//256:7d} {k[2]:6d} {k[3]:3d}+{k[4]:3d} {a['n']:2d} {dur:7.1f} | {wc/waves:9.0f} {pct('SQ_BUSY_CYCLES'):5.0f} {pct('SQ_WAIT_ANY'):6.1f} {pct('SQ_WAIT_INST_ANY'):6.1f} {pct('SQ_ACTIVE_INST_VALU'):6.1f} {mfma_busy:6.1f} | "
          f"{g(a,'SQ_INSTS_VALU')/waves:7.0f} {g(a,'SQ_INSTS_MFMA')/waves:7.0f} {g(a,'SQ_INSTS_LDS')/waves:6.0f} {g(a,'SQ_INSTS_VMEM_RD')/waves:6.0f} {g(a,'SQ_INSTS_SALU')/waves:6.0f} {g(a,'SQ_INSTS_VALU_TRANS_F32')/waves:6.0f} | "
          f"{100*g(a,'SQ_LDS_BANK_CONFLICT')/max(1,g(a,'SQ_LDS_IDX_ACTIVE')):5.1f} | {g(a,'FETCH_SIZE')/1024:7.1f} {g(a,'WRITE_SIZE')/1024:7.1f} {100*g(a,'TCC_HIT')/max(1,g(a,'TCC_HIT')+g(a,'TCC_MISS')):5.1f}")
