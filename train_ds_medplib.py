"""Drop-in name of the reference's training script: `python train_ds_medplib.py --model_size 7b --dataset synthetic ...` (or under
`python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 train_ds_medplib.py ...`).  See medplib_amd/train.py."""
from medplib_amd.train import main

if __name__ == "__main__":
    main()
