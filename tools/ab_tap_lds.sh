for v in notap prod notap prod; do
  echo "== $v"
  if [ $v = prod ]; then unset LORA_HIP_LIB; else export LORA_HIP_LIB=$PWD/build_ab/$v.so; fi
  timeout -k 10 150 python tools/sfd_bench.py 6,2 7,2 8,1 8,2 8,4 2>&1 | grep "^SFD" || exit 1
  timeout -k 10 120 python -c "
import bench, torch, gr_lora_amd as g
print('block leg ms', bench.block_leg(torch, g, torch.device('cuda:0'))['ms'])" 2>&1 | tail -1 || exit 1
done
