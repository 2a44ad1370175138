bash $GRAFT_REPO_ROOT/scripts/_r03_f.sh
bash $GRAFT_REPO_ROOT/scripts/_r03_final.sh
