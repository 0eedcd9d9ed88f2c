for d in 0 1 2 3 4 6 8; do echo "delay $d"; bash profiles/kstats.sh g3w_d$d profiles/default_net_grad_profile_target.py NV=20 G3W_DELAY=$d | grep grad3w | cut -d, -f1-4 | cut -c40-; done
