R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/encpmc2; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT -o l -- python $R/tools/encoder_layers.py once > $OUT/l.log 2>&1; echo "exit $?"
python - <<'PY'
import csv, collections, glob, os
out=os.environ.get("GRAFT_REPO_ROOT", os.getcwd())+"/gpurun_out/encpmc2"
f=glob.glob(out+"/**/l_counter_collection.csv", recursive=True)[0]
by=collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k=(int(r['Dispatch_Id']), r['Kernel_Name'][:48])
    by.setdefault(k,{})[r['Counter_Name']]=float(r['Counter_Value'])
for (d,name),c in list(by.items())[-29:]:
    busy=c.get('SQ_BUSY_CYCLES',1)/32.0
    print("%4d %-46s dur %7.0f cyc  lds_active/CU %.2f  conflict share %.2f  wait_lds %.2f" % (d, name.replace('void hps::',''), busy, c.get('SQ_LDS_IDX_ACTIVE',0)/256/busy, c.get('SQ_LDS_BANK_CONFLICT',0)/max(1,c.get('SQ_LDS_IDX_ACTIVE',1)), c.get('SQ_WAIT_INST_LDS',0)/max(1,c.get('SQ_WAVE_CYCLES',1))))
PY
