O=$GRAFT_REPO_ROOT/gpurun_out/r3v
mkdir -p $O
cd $GRAFT_REPO_ROOT
for rows in 272115 558771; do
for b in x3s_dbg_0 x3s_stag_2 x3s_stag_4 x3s_stag_6 x3s_stag_8 x3s_dbg_0; do echo -n "$b: " >> $O/stagger.txt; timeout -k 5 60 tools/labbin/$b $rows 200 200 >> $O/stagger.txt 2>&1 || exit 1; done; done
cat $O/stagger.txt
