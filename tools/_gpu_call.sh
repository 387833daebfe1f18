cd $GRAFT_REPO_ROOT
O=gpurun_out/r3m26; mkdir -p $O
( echo "task go2_flat (PPO, plane)"; timeout 200 python tools/train_curve.py go2_flat 600 100 2>&1 | grep "^it "; echo; echo "task go2 (PPO, rough curriculum terrain, trimesh walls)"; timeout 200 python tools/train_curve.py go2 600 100 2>&1 | grep "^it "; echo; echo "task go2_cts (CTS, rough)"; timeout 200 python tools/train_curve.py go2_cts 300 100 2>&1 | grep "^it " ) > $O/learning_curves.txt 2>&1
cat $O/learning_curves.txt
