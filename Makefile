# Build of the MI355X SASA engine (gfx950 only).
#   make            -> freesasa_amd/lib/libfreesasa_amd.so (stand-alone drop-in library)
#                      freesasa_amd/lib/libfreesasa_amd_seam.a (seam objects for a drop-in
#                      build of the reference)
#   make emu        -> tests/emu/libsasa_emu.so  (TESTS ONLY: the kernel phase functions
#                      driven on the CPU; never linked into the product)
#   make oracle     -> oracle/ (TESTS ONLY) ; make tools -> tools/libsasa_synth.so
HIPCC   ?= /opt/rocm/bin/hipcc
CC      ?= gcc
CXX     ?= g++
ARCH    ?= gfx950
CSRC     = freesasa_amd/csrc
LIBDIR   = freesasa_amd/lib
HIPFLAGS = --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function
CFLAGS   = -O2 -std=gnu99 -fPIC -ffp-contract=off -Wall

all: $(LIBDIR)/libfreesasa_amd.so $(LIBDIR)/libfreesasa_amd_seam.a

# Device code lives in ONE translation unit (gpu_kernels.hip); the compiler's per-kernel resource report (registers,
# scratch, LDS) is kept next to its object: tests/test_capi.py checks that the hot kernels do not spill.  The other
# .hip files are host code over the HIP runtime (engine_internal.h says who holds what).
ENGINE_HDRS = $(CSRC)/engine_internal.h $(CSRC)/sasa_kernels.h $(CSRC)/sr_caps.h $(CSRC)/lr2_kernels.h $(CSRC)/gpu_parse.h $(CSRC)/protor_table.h include/freesasa_gpu.h include/freesasa_ingest.h
$(LIBDIR)/gpu_kernels.o: $(CSRC)/gpu_kernels.hip $(ENGINE_HDRS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -Rpass-analysis=kernel-resource-usage -c $< -o $@ 2> $(LIBDIR)/kernel_resources.txt; rc=$$?; \
	grep -v "remark:" $(LIBDIR)/kernel_resources.txt >&2; exit $$rc
$(LIBDIR)/gpu_%.o: $(CSRC)/gpu_%.hip $(ENGINE_HDRS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
GPU_OBJS = $(LIBDIR)/gpu_kernels.o $(LIBDIR)/gpu_engine.o $(LIBDIR)/gpu_ops.o $(LIBDIR)/gpu_hostbatch.o $(LIBDIR)/gpu_drivers.o $(LIBDIR)/gpu_parse.o

$(LIBDIR)/seam.o: $(CSRC)/seam.c include/freesasa_amd.h include/freesasa_gpu.h
	@mkdir -p $(LIBDIR)
	$(CC) $(CFLAGS) -c $< -o $@

$(LIBDIR)/testpoints.o: $(CSRC)/testpoints.c include/freesasa_gpu.h
	@mkdir -p $(LIBDIR)
	$(CC) $(CFLAGS) -c $< -o $@

$(LIBDIR)/api.o: $(CSRC)/api.c include/freesasa_amd.h $(CSRC)/hostfault.h
	@mkdir -p $(LIBDIR)
	$(CC) $(CFLAGS) -c $< -o $@

# host-side fault injection (tests): the countdown every allocation / thread creation of the host code asks, and - in
# the shared library ONLY, never in the seam archive that is linked into the reference's build - the library-local
# operator new that puts the engine's C++ allocations behind it (hostfault.h)
$(LIBDIR)/hostfault.o: $(CSRC)/hostfault.c $(CSRC)/hostfault.h
	@mkdir -p $(LIBDIR)
	$(CC) $(CFLAGS) -c $< -o $@
$(LIBDIR)/hostfault_new.o: $(CSRC)/hostfault_new.cpp $(CSRC)/hostfault.h
	@mkdir -p $(LIBDIR)
	$(CXX) -O2 -std=c++17 -fPIC -Wall -c $< -o $@

$(LIBDIR)/ingest.o: $(CSRC)/ingest.c $(CSRC)/protor_table.h include/freesasa_ingest.h $(CSRC)/hostfault.h
	@mkdir -p $(LIBDIR)
	$(CC) $(CFLAGS) -Iinclude -pthread -c $< -o $@

$(LIBDIR)/select.o: $(CSRC)/select.c include/freesasa_ingest.h $(CSRC)/hostfault.h
	@mkdir -p $(LIBDIR)
	$(CC) $(CFLAGS) -Iinclude -c $< -o $@

$(LIBDIR)/ingest_cache.o: $(CSRC)/ingest_cache.c include/freesasa_ingest.h $(CSRC)/hostfault.h
	@mkdir -p $(LIBDIR)
	$(CC) $(CFLAGS) -Iinclude -pthread -c $< -o $@

$(LIBDIR)/libfreesasa_amd.so: $(GPU_OBJS) $(LIBDIR)/seam.o $(LIBDIR)/testpoints.o $(LIBDIR)/api.o $(LIBDIR)/ingest.o $(LIBDIR)/select.o $(LIBDIR)/ingest_cache.o $(LIBDIR)/hostfault.o $(LIBDIR)/hostfault_new.o $(CSRC)/exports.map
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -Wl,--version-script=$(CSRC)/exports.map -o $@ $(filter %.o,$^)

$(LIBDIR)/libfreesasa_amd_seam.a: $(GPU_OBJS) $(LIBDIR)/seam.o $(LIBDIR)/testpoints.o $(LIBDIR)/ingest.o $(LIBDIR)/ingest_cache.o $(LIBDIR)/hostfault.o
	rm -f $@; ar rcs $@ $^

emu: tests/emu/libsasa_emu.so tests/emu/libingest_scalar.so
# the loader with its byte-at-a-time mmCIF tokenizer only: the differential twin of the SSE2 row scanner
tests/emu/libingest_scalar.so: $(CSRC)/ingest.c $(CSRC)/hostfault.c $(CSRC)/hostfault.h $(CSRC)/protor_table.h include/freesasa_ingest.h
	$(CC) $(CFLAGS) -DFREESASA_INGEST_NO_SIMD -Iinclude -pthread -shared -o $@ $(CSRC)/ingest.c $(CSRC)/hostfault.c -lm
tests/emu/libsasa_emu.so: tests/emu/emu.cpp $(CSRC)/sasa_kernels.h $(CSRC)/sr_caps.h $(CSRC)/lr2_kernels.h
	$(CXX) -O2 -std=c++17 -fPIC -ffp-contract=off -DSASA_EMU -shared -o $@ tests/emu/emu.cpp -lm

# Sanitizer build of the HOST sources (the reference's CI runs its C under sanitizers): the parsers,
# the selection language, the C API shims and the test-point generator with AddressSanitizer + UBSan, linked with
# the ordinary (uninstrumented) engine object.  `make asan-test` runs the CPU suites that exercise them.
ASAN_SO = tests/emu/libfreesasa_amd_asan.so
SANFLAGS = -O1 -g -std=gnu99 -fPIC -ffp-contract=off -Wall -fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined
asan: $(ASAN_SO)
$(ASAN_SO): $(CSRC)/api.c $(CSRC)/seam.c $(CSRC)/testpoints.c $(CSRC)/ingest.c $(CSRC)/select.c $(CSRC)/ingest_cache.c $(CSRC)/hostfault.c $(CSRC)/hostfault.h $(CSRC)/protor_table.h $(GPU_OBJS) include/freesasa_amd.h include/freesasa_gpu.h include/freesasa_ingest.h
	for f in api seam testpoints ingest select ingest_cache hostfault; do $(CC) $(SANFLAGS) -Iinclude -pthread -c $(CSRC)/$$f.c -o /tmp/asan_$$f.o || exit 1; done
	$(CXX) -shared -fPIC -o $@ /tmp/asan_api.o /tmp/asan_seam.o /tmp/asan_testpoints.o /tmp/asan_ingest.o /tmp/asan_select.o /tmp/asan_ingest_cache.o /tmp/asan_hostfault.o $(GPU_OBJS) \
	    -fsanitize=address,undefined -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -lamdhip64 -lpthread -lm
asan-test: $(ASAN_SO)
	LD_PRELOAD="$$($(CC) -print-file-name=libasan.so) $$($(CC) -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 \
	    FREESASA_AMD_LIB=$(CURDIR)/$(ASAN_SO) python -m pytest tests/test_ingest.py tests/test_select.py tests/test_capi.py tests/test_hostfault.py -q -m "not gpu" -p no:cacheprovider

oracle: $(LIBDIR)/libfreesasa_amd_seam.a
	$(MAKE) -C oracle all dropin
tools:
	$(MAKE) -C tools

clean:
	rm -rf $(LIBDIR) tests/emu/libsasa_emu.so tests/emu/libingest_scalar.so
	$(MAKE) -C oracle clean
	$(MAKE) -C tools clean
.PHONY: all emu oracle tools clean asan asan-test
