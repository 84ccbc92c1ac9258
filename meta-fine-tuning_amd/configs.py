"""Paths (mirror of the reference's configs.py:1-9; only ``save_dir`` is on the hot path)."""
save_dir = './logs'

miniImageNet_path = 'content/miniImagenet3'
DTD_path = '/ssd/dtd/images/'
ISIC_path = "content"
ChestX_path = "content"
CropDisease_path = "content/CropDiseases"
EuroSAT_path = "content/2750"
