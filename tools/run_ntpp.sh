B=tools/diag/bin/nt_pp
for shape in "16384 1536 512" "16384 512 512" "16384 1024 512" "16384 512 1024" "16384 512 1536" "10368 1536 512" "10368 512 1024" "16384 512 4096"; do
  for v in 0 1 2; do timeout -k 5 60 $B $shape $v 0 || echo "variant $v FAILED rc $?"; done
done
timeout -k 5 60 $B 16384 512 1024 1 1
timeout -k 5 60 $B 16384 512 1024 2 1
