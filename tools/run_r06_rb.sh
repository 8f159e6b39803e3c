R=$GRAFT_REPO_ROOT
cd $R
for v in ${1:-hip exp_rb_nostore exp_rb_nogelu}; do echo "== $v"; ADT_LIB_PATH=$R/adt_str_amd/libadt_$v.so timeout -k 10 120 python tools/probe/rowblock384.py 2>&1 | grep -v "amdgpu.ids\|ADT_LIB_PATH"; done
