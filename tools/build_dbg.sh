# Debug build of the library (timing switches compiled in) beside the product build:
# adell_mri_amd/libadellhip_dbg.so, objects under /tmp/adell_dbg. Usage: bash tools/build_dbg.sh
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
B=/tmp/adell_dbg
mkdir -p $B/adell_mri_amd $B/include
rsync -a --exclude '*.o' --exclude '*.ru.txt' $ROOT/adell_mri_amd/csrc $B/adell_mri_amd/ 2>/dev/null || cp -r $ROOT/adell_mri_amd/csrc $B/adell_mri_amd/
cp $ROOT/include/*.h $B/include/
make -C $B/adell_mri_amd/csrc -j8 ADELL_DEBUG=1 OUT=$ROOT/adell_mri_amd/libadellhip_dbg.so | tail -1
