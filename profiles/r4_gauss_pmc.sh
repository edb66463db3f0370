cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 5 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_GF -o pmc -- python3 profiles/gauss_time.py > /dev/null 2>&1
timeout -k 5 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_GW -o pmc -- python3 profiles/gauss_time.py > /dev/null 2>&1
python3 profiles/pmc_summary.py gpurun_out/pmc_GF gpurun_out/pmc_GW gpurun_out/r04_gauss_pmc_traffic.json | grep -i gauss
rm -rf gpurun_out/pmc_GF gpurun_out/pmc_GW
