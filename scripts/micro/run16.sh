O=gpurun_out/run16; mkdir -p $O; B=scripts/micro/_bin
for P in 0 1; do
  echo "== FRCNN_PRIO=$P" >> $O/time.txt
  FRCNN_PRIO=$P $B/conv_lab time all 0 >> $O/time.txt 2>&1
done
for P in 0 1; do echo "== FRCNN_PRIO=$P" >> $O/stamps.txt; FRCNN_PRIO=$P $B/conv_lab_stamps stamps s5_2c 0 >> $O/stamps.txt 2>&1; FRCNN_PRIO=$P $B/conv_lab_stamps stamps s4_2c 0 >> $O/stamps.txt 2>&1; done
# split-K factor sweep on the trunk shapes with long k loops, and the stem with the variant-2 loop
$B/conv_lab time trunk 223,323,423,523,623,823 s4_2b > $O/splits.txt 2>&1
$B/conv_lab time trunk 223,323,423,523,623,823 s3_2b >> $O/splits.txt 2>&1
$B/conv_lab time trunk 123,223,323,423,523,623 s4x_2a >> $O/splits.txt 2>&1
$B/conv_lab time trunk 123,223,323,423,523,623 s3x_2a >> $O/splits.txt 2>&1
$B/conv_lab time trunk 223,323,423,523,623,823 rpn_conv1 >> $O/splits.txt 2>&1
$B/conv_lab time trunk 0,31 conv1 >> $O/splits.txt 2>&1
grep -E "==|total" $O/time.txt; grep -E "==|tile|prologue|main loop|epilogue issue|lifetime" $O/stamps.txt | grep -B1 -A4 "rep 1"; grep -vE "^layer|^total" $O/splits.txt
