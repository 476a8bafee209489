for s in 0 -1; do python tools/upload_prof.py --schedule $s; YDS_UPLOAD_AFTER_PASS=1 python tools/upload_prof.py --schedule $s;  done
for s in 0 -1; do python tools/upload_prof.py --schedule $s; YDS_UPLOAD_AFTER_PASS=1 python tools/upload_prof.py --schedule $s;  done
