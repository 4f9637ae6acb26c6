#!/bin/bash
# usage: gpu_f32_blocks.sh <out-prefix>   (through gpurun) -- block size / waves sweep of the fp32 per-cell geometry kernels
# (fp64 block accumulator: 4 more bytes of LDS per local dof than in round 2), one line per run.
out=$1
lib=fenicsx-fus_amd/fenicsxfus_amd/libfusmi.so
for geom in trilinear auto; do
  tools/ab_matrix.sh ${out}_p4_$geom 1 "--dtype f32 --P 4 --geometry $geom" -- $lib -- "--block-elems 16 --waves 4" "--block-elems 24 --waves 4" "--block-elems 32 --waves 4" "--block-elems 32 --waves 8" "--block-elems 48 --waves 8" "--block-elems 64 --waves 8"
  tools/ab_matrix.sh ${out}_p5_$geom 1 "--dtype f32 --P 5 --geometry $geom" -- $lib -- "--block-elems 8" "--block-elems 12" "--block-elems 16" "--block-elems 24"
  tools/ab_matrix.sh ${out}_p6_$geom 1 "--dtype f32 --P 6 --geometry $geom" -- $lib -- "--block-elems 4" "--block-elems 8" "--block-elems 12" "--block-elems 16"
  tools/ab_matrix.sh ${out}_p7_$geom 1 "--dtype f32 --P 7 --geometry $geom" -- $lib -- "--block-elems 4" "--block-elems 8" "--block-elems 12" "--block-elems 16"
done
cat gpurun_out/${out}_p*_*.txt > gpurun_out/${out}_all.txt
