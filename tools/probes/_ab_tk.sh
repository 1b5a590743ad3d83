L=transformer-inertial-poser_amd/csrc/libtip_hip.so
for v in old prev new; do cp tools/probes/_$v.so $L; echo "== $v"; bash tools/train_kernels.sh 2>/dev/null | grep "bwd_kernel\|encoder_h\|dwgemm"; done
cp tools/probes/_new.so $L
