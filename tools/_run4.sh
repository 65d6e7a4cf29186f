mkdir -p gpurun_out/r02d
python tools/rotation_roundtrip_probe.py > gpurun_out/r02d/rt_res.json 2> gpurun_out/r02d/rt.err
SCRI_AMD_ROTATE_STAGED=1 python tools/rotation_roundtrip_probe.py > gpurun_out/r02d/rt_staged.json 2>> gpurun_out/r02d/rt.err
cat gpurun_out/r02d/rt_*.json; tail -3 gpurun_out/r02d/rt.err
