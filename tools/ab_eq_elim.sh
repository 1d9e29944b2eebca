# same-box A/B of the equality-row elimination (MPC_NO_EQ_ELIM=1 switches it off) on the configurations with equality rows
mkdir -p gpurun_out/r3
for w in c2 c2x20; do
  for e in 0 1; do echo "== $w MPC_NO_EQ_ELIM=$e"; MPC_NO_EQ_ELIM=$e timeout 300 python tools/ab_lib.py ppopt_amd/csrc/libmpcombi_hip.so $w 2>&1 | tail -1; done
done
