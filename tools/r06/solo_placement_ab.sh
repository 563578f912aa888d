R=${GRAFT_REPO_ROOT:-$(pwd)}
for t in . _b5 . _b5; do cd $R/$t; echo "== $t"; timeout -k 10 120 python $R/tools/probes/dbg_solo_placement.py 2>&1 | grep -v amdgpu.ids | cut -c1-500; done
