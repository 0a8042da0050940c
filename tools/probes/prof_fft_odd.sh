export TMPDIR=/tmp
mkdir -p gpurun_out/r04
out=gpurun_out/r04/odd_prof
rm -rf $out
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/run_fft_odd_shapes.py 0 > $out.out 2> $out.err < /dev/null
python3 - <<'PY'
import csv,glob
f=sorted(glob.glob('gpurun_out/r04/odd_prof/*/*kernel_trace.csv'))[-1]
rows=[r for r in csv.DictReader(open(f)) if 'fft_lines' in r['Kernel_Name'] or 'fluid_kernel' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# first 7 after warmup: take rows 7*3 .. 7*4 (mode 3 ran first: warm 3 + reps 10)
seq=[(r['Kernel_Name'][:30], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))//1000, r.get('LDS_Block_Size'), int(r['Grid_Size_X'])//256) for r in rows[21:28]]
for x in seq: print(x)
PY
