ulimit -c 0
for cfg in "PXR_UPLOAD_NO_STREAMING=1 PXR_UPLOAD_THREADS=32" "PXR_UPLOAD_THREADS=32" "PXR_UPLOAD_THREADS=16" "PXR_UPLOAD_THREADS=8" "PXR_UPLOAD_NO_STREAMING=1 PXR_UPLOAD_THREADS=8"; do
echo "== $cfg"; env $cfg timeout 100 python tools/_upload_probe.py 2>&1 | grep "GB/s" | tail -3
done
