# build an experimental variant of the library: tools/build_variant.sh NAME [-DFLAG ...]
NAME=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -c freesasa_amd/csrc/gpu_kernels.hip -o /tmp/ge_$NAME.o && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=freesasa_amd/csrc/exports.map -o freesasa_amd/lib/libvar_$NAME.so /tmp/ge_$NAME.o freesasa_amd/lib/gpu_engine.o freesasa_amd/lib/gpu_ops.o freesasa_amd/lib/gpu_hostbatch.o freesasa_amd/lib/gpu_drivers.o freesasa_amd/lib/ingest_cache.o freesasa_amd/lib/select.o freesasa_amd/lib/seam.o freesasa_amd/lib/testpoints.o freesasa_amd/lib/api.o freesasa_amd/lib/ingest.o freesasa_amd/lib/hostfault.o freesasa_amd/lib/hostfault_new.o && echo built libvar_$NAME.so
