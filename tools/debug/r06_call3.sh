cd $GRAFT_REPO_ROOT
for A in 0 1 2 3 4 8 12 15; do echo "ablate $A"; HHSR_FFTW_ABLATE=$A timeout 200 python tools/fft_ab.py 3000 4000 1 2>&1 | grep WAVE; done
