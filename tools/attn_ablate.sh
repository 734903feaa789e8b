set -e
cd $GRAFT_REPO_ROOT
python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids | sed 's/^/BASE  /'
for a in 1 2 3; do
  MMD_EXTRA_CXXFLAGS="-DATTN_ABLATE=$a" python mm-diffusion_amd/build.py --force > /dev/null 2>&1
  python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids | head -3 | sed "s/^/ABL$a  /"
done
