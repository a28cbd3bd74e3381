"""Copy the evidence of tools/profile_r06.sh from gpurun_out/r06/ into profiles/ (tag r06) and rebuild profiles/pmc_traffic.json."""
import csv, glob, json, os, shutil
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out, prof, tag = os.path.join(root, 'gpurun_out', 'r06'), os.path.join(root, 'profiles'), 'r06'
# detail records (every leg of a run) as profiles/r05_bench_<w>.json, the bounded contract lines as printed as profiles/r05_bench_<w>_line.json
for f in glob.glob(os.path.join(out, 'bench_*.json')) + glob.glob(os.path.join(out, 'force_shard_*.json')) + glob.glob(os.path.join(out, 'mb_dimshard.json')) + glob.glob(os.path.join(out, 'mb_idpath.json')):
    name = os.path.basename(f)
    if 'pmc_' in name or not os.path.getsize(f):
        continue
    dst = os.path.join(prof, f'{tag}_{name}')
    if name.endswith('_line.json'):
        open(dst, 'w').write(open(f).read().strip().splitlines()[-1] + '\n')
    else:
        try:
            json.dump(json.load(open(f)), open(dst, 'w'), indent=None, separators=(',', ':'))
            open(dst, 'a').write('\n')
        except ValueError:                      # (--only-e2e / --only-ingest print one JSON line behind library banners)
            open(dst, 'w').write(open(f).read().strip().splitlines()[-1] + '\n')
for w in ('c5_headline', 'c1', 'c2', 'c3', 'c4', 'fullsort_conet', 'force_shard_row', 'ingest'):
    ks = glob.glob(os.path.join(out, f'trace_{w}', '**', '*kernel_stats.csv'), recursive=True)
    if ks:
        shutil.copy(ks[0], os.path.join(prof, f'{tag}_{w}_kernel_stats.csv' if w in ('force_shard_row', 'ingest') else f'{tag}_bench_{w}_kernel_stats.csv'))
for t in ('mb_fullsort_d128', 'mb_fullsort_d64', 'mb_fullsort_topk_u256', 'mb_fullsort_topk_u1024', 'mb_models5', 'mb_mapstep', 'gputests'):
    f = os.path.join(out, t + '.txt')
    if os.path.exists(f):
        txt = [l for l in open(f).read().splitlines() if 'amdgpu.ids' not in l]
        open(os.path.join(prof, f'{tag}_{t}.txt'), 'w').write('\n'.join(txt) + '\n')
# ---- PMC: mean FETCH_SIZE / WRITE_SIZE per kernel over the headline's dispatches
short = lambda n: n.replace('(anonymous namespace)::', '').replace('void ', '', 1).split('(')[0]
raw, order = {}, {}
for C in ('FETCH_SIZE', 'WRITE_SIZE'):
    src = glob.glob(os.path.join(out, 'pmc_' + C, '**', '*counter_collection.csv'), recursive=True)
    if not src:
        continue
    rows = sorted([r for r in csv.DictReader(open(src[0])) if r['Counter_Name'] == C], key=lambda r: int(r['Dispatch_Id']))
    by = {}
    for r in rows:
        by.setdefault(short(r['Kernel_Name']), []).append(float(r['Counter_Value']))
    HEAD = ('bpr_fwd_apply_kernel<32', 'batch_norms_kernel<32', 'occ_flags_kernel', 'make_keys2_kernel', 'rowwise_apply_dups2_kernel<32, 1>')
    for k, vals in by.items():
        big = [v for v in vals if v > 0.5 * max(vals)] if max(vals) > 0 else vals
        if any(h in k for h in HEAD):
            big = vals[:8]          # dispatch order: (1 warm-up + 3 timed steps) x 2 domains of the HEADLINE batch come first; later legs
                                    # (k = 4 comparison, synthetic grid) launch the same kernels on other batches
        raw.setdefault(k, {})[C + '_KiB'] = round(sum(big) / len(big), 2)
        raw[k]['dispatches_' + C] = len(big)
        raw[k]['all_dispatches'] = len(vals)
        raw[k]['sum_' + C + '_KiB'] = round(sum(vals), 2)
        raw[k]['head_sum_' + C + '_KiB'] = round(sum(vals[:8]), 2)
    keep = [r for r in rows if any(s in r['Kernel_Name'] for s in ('rowwise_apply', 'bpr_fwd', 'batch_norms', 'occ_flags', 'apply2', 'map_step', 'map_pipe', 'seg_piece', 'seg_long', 'radix_sort', 'onesweep', 'make_keys', 'rank_'))]
    with open(os.path.join(prof, f'{tag}_pmc_{C}_dispatches.csv'), 'w', newline='') as fh:
        w = csv.DictWriter(fh, fieldnames=['Dispatch_Id', 'Grid_Size', 'Kernel_Name', 'Counter_Name', 'Counter_Value'])
        w.writeheader()
        for r in keep[:4000]:
            r2 = {k: r[k] for k in w.fieldnames}
            r2['Kernel_Name'] = short(r2['Kernel_Name'])[:90]
            w.writerow(r2)
byt = lambda k: int(raw[k].get('FETCH_SIZE_KiB', 0) * 1024 * 2 + raw[k].get('WRITE_SIZE_KiB', 0) * 1024) if k in raw else None
find = lambda pat: next((k for k in raw if pat in k), None)
names = {'bpr_fwd_apply_kernel': find('bpr_fwd_apply_kernel<32'), 'batch_norms_kernel': find('batch_norms_kernel<32'), 'occ_flags_kernel': find('occ_flags_kernel'),
         'rowwise_apply_dups2_kernel': find('rowwise_apply_dups2_kernel<32, 1>'),        # both tables' duplicate rows in one launch (round 4)
         'bpr_fwd_kernel': find('bpr_fwd_kernel<32'), 'bpr_fwd_kmajor_kernel(k=4)': find('bpr_fwd_kmajor_kernel<32, 4, 2>'),
         'map_step_kernel': find('map_pipe_kernel') or find('map_step_kernel')}
o = {'_note': 'HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes of `python bench.py --no-cpu-baseline --no-fullsort '
              '--no-config-legs --no-e2e --no-ingest --single-stream --steps 3 --warmup 1`, tools/profile_r06.sh; B = 1,048,576 triples per domain, D = 128, row-wise Adam). FETCH_SIZE x 1024 x 2 '
              '(gfx950 wide-stream correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE x 1024. Mean over each kernel\'s first 8 dispatches = the '
              'headline batch (1 warm-up + 3 timed steps x 2 domains). `domain_step` = batch norms + sort + flags + forward/optimizer + the duplicate-row '
              'apply of both tables (the sort\'s share: all make_keys / rocPRIM dispatches of the run divided by its number of sorts).',
     '_raw': {v: raw[v] for v in names.values() if v}}
for k, v in names.items():
    if v:
        o[k] = byt(v)
sort_k = [k for k in raw if 'radix_sort' in k or 'make_keys2' in k or 'onesweep' in k or 'merge_sort' in k]
n_sorts = raw.get(find('make_keys2_kernel') or '', {}).get('all_dispatches', 0)
need = ('bpr_fwd_apply_kernel', 'batch_norms_kernel', 'occ_flags_kernel', 'rowwise_apply_dups2_kernel')
if n_sorts and all(names[k] for k in need):
    sort_bytes = sum(raw[k].get('sum_FETCH_SIZE_KiB', 0) * 2048 + raw[k].get('sum_WRITE_SIZE_KiB', 0) * 1024 for k in sort_k) / n_sorts
    o['sort_ids(per call)'] = int(sort_bytes)
    o['domain_step'] = int(sum(o[k] for k in need) + sort_bytes)
import datetime
o['_collected'] = datetime.date.today().isoformat()
o['_round'] = 6
json.dump(o, open(os.path.join(prof, 'pmc_traffic.json'), 'w'), indent=1)
print({k: v for k, v in o.items() if not k.startswith('_')})
# ---- MFMA utilisation of the U = 1,024 scoring kernels of THIS build (plain scoring D = 128 / 64, fused mask + top-10)
score = {}
for name, D, pat in (('score_d128', 128, 'score_persistent_kernel'), ('score_d64', 64, 'score_persistent_kernel'), ('score_topk', 128, 'score_persistent_kernel')):
    src = glob.glob(os.path.join(out, 'pmc_' + name, '**', '*counter_collection.csv'), recursive=True)
    if not src:
        continue
    by = {}
    for r_ in csv.DictReader(open(src[0])):
        if pat in r_['Kernel_Name']:
            by.setdefault(short(r_['Kernel_Name']), {}).setdefault(r_['Counter_Name'], []).append(float(r_['Counter_Value']))
    for k, acc in by.items():
        if acc.get('SQ_VALU_MFMA_BUSY_CYCLES') and acc.get('GRBM_GUI_ACTIVE'):
            busy = sum(acc['SQ_VALU_MFMA_BUSY_CYCLES']) / len(acc['SQ_VALU_MFMA_BUSY_CYCLES'])
            act = sum(acc['GRBM_GUI_ACTIVE']) / len(acc['GRBM_GUI_ACTIVE'])
            N, U = 10_000_001, 1024
            mfma = 2.0 * U * N * D / 4096.0                       # v_mfma_f32_32x32x2_f32 = 4,096 flop
            score[name + ': ' + k] = {'dispatches': len(acc['GRBM_GUI_ACTIVE']), 'counters_mean': {'SQ_VALU_MFMA_BUSY_CYCLES': busy, 'GRBM_GUI_ACTIVE': act},
                                      'mfma_instructions': mfma, 'active_cycles_per_xcd': act / 8.0,
                                      'mfma_pipe_utilisation_from_instruction_count': mfma * 64.0 / ((act / 8.0) * 1024.0),
                                      'mfma_busy_counter_over_active_cycles_x_simds': busy / ((act / 8.0) * 1024.0)}
if score:
    score['_note'] = ('MB_U=1024 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- python tools/mb_fullsort.py {128,64} / tools/mb_fullsort_topk.py 128 '
                      '(N = 10,000,001), own passes, round 6 build.  GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1,024 SIMDs; a v_mfma_f32_32x32x2_f32 occupies its SIMD\'s matrix '
                      'pipe for 64 cycles.  The top-k pass also launches the plain kernel (the A/B in the script): both are listed.')
    json.dump(score, open(os.path.join(prof, f'{tag}_pmc_score_mfma.json'), 'w'), indent=1)
    print({k: round(v['mfma_busy_counter_over_active_cycles_x_simds'], 3) for k, v in score.items() if not k.startswith('_')})

# ---- matrix-pipe utilisation of the mapping and CoNet kernels (own --pmc passes over tools/mb_conet.py / mb_mapstep.py)
mf = {}
for T in ('conet', 'c3', 'mapstep', 'map'):
    src = glob.glob(os.path.join(out, 'pmc_mfma_' + T, '**', '*counter_collection.csv'), recursive=True)
    if not src:
        continue
    by = {}
    for r in csv.DictReader(open(src[0])):
        by.setdefault(short(r['Kernel_Name']), {}).setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
    for k, v in by.items():
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in v and 'GRBM_GUI_ACTIVE' in v and any(x in k for x in ('conet_fb', 'conet_wgrad_kernel', 'conet_fullsort_kernel', 'map_pipe')) and 'finish' not in k:
            b, a = v['SQ_VALU_MFMA_BUSY_CYCLES'], v['GRBM_GUI_ACTIVE']
            n = min(len(a), len(b))
            b, a = b[n // 4:n], a[n // 4:n]
            mb, ma = sum(b) / len(b), sum(a) / len(a)
            mf[k] = {'dispatches': len(b), 'SQ_VALU_MFMA_BUSY_CYCLES': mb, 'GRBM_GUI_ACTIVE': ma, 'mfma_busy_over_active_cycles_x_simds': mb / ((ma / 8) * 1024)}
if mf:
    note = ("rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace over tools/mb_conet.py (C3 shape, 8,190 rows) and tools/mb_mapstep.py "
            "(OVERLAP step, linear and tanh-MLP mappings, OB = 100 and 65,536: the averages are dominated by the OB = 65,536 launches) and over `bench.py --workload c3` "
            "(the CoNet full-sort leg: conet_fullsort_kernel at its four shapes, averaged), own passes, one MI355X, round 6.  GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1,024 SIMDs; mfma_busy_over_active_cycles_x_simds = share of the kernel's cycles in which a "
            "SIMD's matrix pipe is busy. First quarter of each kernel's dispatches (warm-up) dropped.")
    json.dump(dict({'_note': note}, **mf), open(os.path.join(prof, f'{tag}_pmc_mfma_conet_map.json'), 'w'), indent=1)
