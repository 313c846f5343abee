set -e
for v in ""; do
  touch gnn_hex_amd/csrc/qnet_fused_kernels.h
  make -C gnn_hex_amd/csrc STAMPS=1 EXTRA="$v" 2>&1 | grep -i "error" || true
  echo "=== variant [$v]"
  timeout -k 10 200 python tools/stamps.py 2>/dev/null | grep -v "loads-out\|reads-in\|M1 "
done
