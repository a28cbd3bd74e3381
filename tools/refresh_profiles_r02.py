"""Copy the evidence of tools/profile_r02.sh from gpurun_out/r02/ into profiles/ (tag r02) and rebuild profiles/pmc_traffic.json."""
import csv, glob, json, os, shutil
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out, prof, tag = os.path.join(root, 'gpurun_out', 'r02'), os.path.join(root, 'profiles'), 'r02'
last = lambda f: open(f).read().strip().splitlines()[-1] + '\n'
for w in ('c1', 'c2', 'c3', 'c4', 'c5', 'c5_200steps', 'c3_dense_adam', 'c4_full_last_layer', 'c5_under_rocprof', 'c3_under_rocprof'):
    f = os.path.join(out, f'bench_{w}.json')
    if os.path.exists(f):
        open(os.path.join(prof, f'{tag}_bench_{w}.json'), 'w').write(last(f))
for w in ('c5', 'c3'):
    ks = glob.glob(os.path.join(out, f'trace_{w}', '**', '*kernel_stats.csv'), recursive=True)
    if ks:
        shutil.copy(ks[0], os.path.join(prof, f'{tag}_bench_{w}_kernel_stats.csv'))
for t in ('sweep_small', 'mb_kmajor', 'mb_mapstep', 'mb_conet', 'mb_smallsort', 'mb_models5'):
    f = os.path.join(out, t + '.txt')
    if os.path.exists(f):
        txt = [l for l in open(f).read().splitlines() if 'amdgpu.ids' not in l and 'radix sort) 0.0000' not in l]
        open(os.path.join(prof, f'{tag}_{t}.txt'), 'w').write('\n'.join(txt) + '\n')
# ---- PMC: mean FETCH_SIZE / WRITE_SIZE per kernel over the step's own (large) dispatches
short = lambda n: n.replace('(anonymous namespace)::', '').replace('void ', '', 1).split('(')[0]
raw = {}
for C in ('FETCH_SIZE', 'WRITE_SIZE'):
    src = glob.glob(os.path.join(out, 'pmc_' + C, '**', '*counter_collection.csv'), recursive=True)
    if not src:
        continue
    rows = sorted([r for r in csv.DictReader(open(src[0])) if r['Counter_Name'] == C], key=lambda r: int(r['Dispatch_Id']))
    by = {}
    for r in rows:
        by.setdefault(short(r['Kernel_Name']), []).append(float(r['Counter_Value']))
    HEAD = ('bpr_fwd_grad_kernel<32, false>', 'rowwise_apply_kernel<32, 1, false>', 'rowwise_apply_kernel<32, 1, true>')
    for k, vals in by.items():
        big = [v for v in vals if v > 0.5 * max(vals)] if max(vals) > 0 else vals
        if any(h in k for h in HEAD):
            big = vals[:8]          # dispatch order: (1 warm-up + 3 timed steps) x 2 domains of the HEADLINE batch come first; the legs
                                    # that follow (k = 4 comparison, OVERLAP) launch the same kernels on other batches
        raw.setdefault(k, {})[C + '_KiB'] = round(sum(big) / len(big), 2)
        raw[k]['dispatches_' + C] = len(big)
        raw[k]['all_dispatches'] = len(vals)
        raw[k]['sum_' + C + '_KiB'] = round(sum(vals), 2)
    keep = [r for r in rows if any(s in r['Kernel_Name'] for s in ('rowwise_apply', 'bpr_fwd', 'apply2', 'map_step', 'map_pipe', 'seg_piece', 'seg_long', 'radix_sort', 'make_keys', 'rank_'))]
    with open(os.path.join(prof, f'{tag}_pmc_{C}_dispatches.csv'), 'w', newline='') as fh:
        w = csv.DictWriter(fh, fieldnames=['Dispatch_Id', 'Grid_Size', 'Kernel_Name', 'Counter_Name', 'Counter_Value'])
        w.writeheader()
        for r in keep:
            r2 = {k: r[k] for k in w.fieldnames}
            r2['Kernel_Name'] = short(r2['Kernel_Name'])[:90]
            w.writerow(r2)
byt = lambda k: int(raw[k].get('FETCH_SIZE_KiB', 0) * 1024 * 2 + raw[k].get('WRITE_SIZE_KiB', 0) * 1024) if k in raw else None
find = lambda pat: next((k for k in raw if pat in k), None)
names = {'bpr_fwd_grad_kernel': find('bpr_fwd_grad_kernel<32, false>'), 'rowwise_apply_kernel(users)': find('rowwise_apply_kernel<32, 1, false>'),
         'rowwise_apply_kernel(items)': find('rowwise_apply_kernel<32, 1, true>'), 'bpr_fwd_kmajor_kernel(k=4)': find('bpr_fwd_kmajor_kernel<32, 4, 2>'),
         'apply2_kernel(items, coefficient x user row)': find('apply2_kernel<32, 1, 1>'), 'apply2_kernel(users, S rows)': find('apply2_kernel<32, 1, 0>'),
         'map_step_kernel': find('map_pipe_kernel') or find('map_step_kernel')}   # the OVERLAP step's first launch, whichever form ran
o = {'_note': 'HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes of `python bench.py --no-cpu-baseline --no-fullsort '
              '--steps 3 --warmup 1`, tools/profile_r02.sh; B = 1,048,576 triples per domain, D = 128, row-wise Adam). FETCH_SIZE x 1024 x 2 (gfx950 '
              'wide-stream correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE x 1024. Mean over each kernel\'s large dispatches (the small-batch '
              'and OVERLAP legs launch some of the same kernels on far fewer rows: those dispatches are left out). `domain_step` = forward + '
              'both applies + the radix sort\'s share (all make_keys / rocPRIM sort dispatches of the run divided by its number of sorts).',
     '_raw': {v: raw[v] for v in names.values() if v}}
for k, v in names.items():
    if v:
        o[k] = byt(v)
sort_k = [k for k in raw if 'radix_sort' in k or 'make_keys2' in k or 'onesweep' in k or 'merge_sort' in k]
n_sorts = raw.get(find('make_keys2_kernel') or '', {}).get('all_dispatches', 0)
if n_sorts and all(names[k] for k in ('bpr_fwd_grad_kernel', 'rowwise_apply_kernel(users)', 'rowwise_apply_kernel(items)')):
    sort_bytes = sum(raw[k].get('sum_FETCH_SIZE_KiB', 0) * 2048 + raw[k].get('sum_WRITE_SIZE_KiB', 0) * 1024 for k in sort_k) / n_sorts
    o['sort_ids(per call)'] = int(sort_bytes)
    o['domain_step'] = int(o['bpr_fwd_grad_kernel'] + o['rowwise_apply_kernel(users)'] + o['rowwise_apply_kernel(items)'] + sort_bytes)
json.dump(o, open(os.path.join(prof, 'pmc_traffic.json'), 'w'), indent=1)
print({k: v for k, v in o.items() if not k.startswith('_')})
