# Ablation builds of the MFMA attention kernel (results are WRONG by construction; timing only).
# 1: no exp  2: no V^T LDS writes  3: no PV MFMAs  4: no K/V global loads after tile 0  5: no LDS staging after tile 0
set -e
cd ${GRAFT_REPO_ROOT:-.}
python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids | head -3 | sed 's/^/BASE  /'
for a in ${ABLATIONS:-1 2 3 4 5}; do
  MMD_EXTRA_CXXFLAGS="-DATTN_ABLATE=$a" python mm-diffusion_amd/build.py --force > /dev/null 2>&1
  python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids | head -3 | sed "s/^/ABL$a  /"
done
python mm-diffusion_amd/build.py --force > /dev/null 2>&1
