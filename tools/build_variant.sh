# build an experimental variant of the library: tools/build_variant.sh NAME [-DFLAG ...]
NAME=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -c freesasa_amd/csrc/gpu_engine.hip -o /tmp/ge_$NAME.o && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o freesasa_amd/lib/libvar_$NAME.so /tmp/ge_$NAME.o freesasa_amd/lib/seam.o freesasa_amd/lib/testpoints.o freesasa_amd/lib/api.o freesasa_amd/lib/ingest.o && echo built libvar_$NAME.so
