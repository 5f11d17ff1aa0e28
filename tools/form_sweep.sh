# frame times of the regen kernel's loop forms (default scene, 64 spp) over image sizes
for size in "1280 720" "1920 1080" "2560 1440" "3840 2160"; do
  for cfg in plain: batched: mixed:16 mixed:32 mixed:64; do
    form=${cfg%%:*}; div=${cfg#*:}
    printf "%s %s div=%s: " "$size" $form "$div"
    RT_REGEN_FORM=$form RT_REGEN_CHAIN_DIV=${div:-32} timeout -k 10 100 python tools/ab_lib.py $size | cut -c1-40 || exit 1
  done
done
