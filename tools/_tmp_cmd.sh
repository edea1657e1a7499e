mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1; REPO=$PWD
for ag in 1 2 4 8 16 37; do echo "agroup $ag"; python tools/kbench.py --only tdnn2,tdnn5,lstm_proj --agroup $ag 2>&1 | grep "_split  "; done
