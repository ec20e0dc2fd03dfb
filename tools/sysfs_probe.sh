echo "== sysfs probe"; ls /sys/class/drm/ 2>&1 | head; for c in /sys/class/drm/card*/device; do echo $c; readlink -f $c; cat $c/pp_dpm_sclk 2>&1 | head -5; ls $c/hwmon/*/ 2>&1 | tr '\n' ' ' | head -c 600; echo; cat $c/hwmon/*/power1_average $c/hwmon/*/power1_input 2>&1 | head -3; done
python - <<'PY'
import torch
p=torch.cuda.get_device_properties(0)
print({k:getattr(p,k) for k in dir(p) if not k.startswith('_') and k in ('pci_bus_id','pci_device_id','pci_domain_id','multi_processor_count','name')})
PY
