O=$GRAFT_REPO_ROOT/gpurun_out/r3f
mkdir -p $O
cd $GRAFT_REPO_ROOT
for d in 0 1 2 4 8 16 32 3 7 23 19 17 0; do timeout -k 5 60 tools/labbin/x3s_dbg_$d 272115 200 200 >> $O/x3s_dbg.txt 2>&1 || break; done
for d in 0 1 17 23; do timeout -k 5 60 tools/labbin/x3s_dbg_$d 558771 200 200 >> $O/x3s_dbg.txt 2>&1 || break; done
cat $O/x3s_dbg.txt
