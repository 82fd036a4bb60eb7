# host-side scaling of the PDB / mmCIF loader on the GPU box's cores (no GPU work)
gcc -O2 -pthread -Iinclude -Ifreesasa_amd/csrc tools/dev/ingest_scaling_batch.c -o /tmp/isb
nproc
for t in 1 8 16 32 64 128; do FREESASA_INGEST_TIMING=1 /tmp/isb $t ${1:-32} 2>&1 | tail -1; done
for t in 1 8 16 32 64; do /tmp/isb $t ${1:-32} cif 2>&1 | tail -1; done
