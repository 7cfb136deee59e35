cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
sed -i "s/names = {0: 'full', 4: 'no gemm', 32: 'no valu role', 36: 'neither'} if os.environ.get('DUAL')/names = {0: 'full'} if os.environ.get('ONLYFULL') else {0: 'full', 4: 'no gemm', 32: 'no valu role', 36: 'neither'} if os.environ.get('DUAL')/" tools_ablate.py
for dual in 1 0; do
ONLYFULL=1 CMDGEN_EDGE_DUAL=$dual rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d gpurun_out/pmc_d$dual -- python3 tools_ablate.py 16 full-atom > gpurun_out/pmc_d$dual.log 2>&1
ONLYFULL=1 CMDGEN_EDGE_DUAL=$dual rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD --output-format csv -d gpurun_out/pmc_e$dual -- python3 tools_ablate.py 16 full-atom > gpurun_out/pmc_e$dual.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for d in ('pmc_d1','pmc_e1','pmc_d0','pmc_e0'):
    fs = glob.glob(f'gpurun_out/{d}/*/*counter_collection.csv')
    if not fs: print('no file', d); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda:[0.0,0]))
    for row in csv.DictReader(open(fs[0])):
        k = row['Kernel_Name'].split('(')[0][-30:]
        agg[k][row['Counter_Name']][0] += float(row['Counter_Value']); agg[k][row['Counter_Name']][1] += 1
    for k in agg:
        if 'edge_msg' in k:
            print(d, k, {c: round(v[0]/v[1]) for c, v in agg[k].items()}, 'n', list(agg[k].values())[0][1])
PY
