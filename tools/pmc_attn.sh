cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d /tmp/pa -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for f in glob.glob('/tmp/pa/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:40]
        if 'attn' in k or 'ln_' in k:
            agg[k][r['Counter_Name']]+=float(r['Counter_Value']); 
for k,a in agg.items():
    print(k, {c:'%.3g'%v for c,v in a.items()})
    print('   bank conflict / idx active = %.2f ; wait_any %.0f%% wait_inst %.0f%% active %.0f%%' % (a['SQ_LDS_BANK_CONFLICT']/max(a['SQ_LDS_IDX_ACTIVE'],1), 100*a['SQ_WAIT_ANY']/a['SQ_WAVE_CYCLES'], 100*a['SQ_WAIT_INST_ANY']/a['SQ_WAVE_CYCLES'], 100*a['SQ_ACTIVE_INST_ANY']/a['SQ_WAVE_CYCLES']))
PY
