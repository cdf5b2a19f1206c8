"""Drop-in for the reference's data/LRHR_dataset.py (same constructor, same __len__, same directory / lmdb key
layout, data/LRHR_dataset.py:9-99) re-designed for a device-side transform: __getitem__ only DECODES -- it returns
uint8 HWC tensors plus the sample's horizontal-flip draw -- and the float conversion / flip / range map of
transform_augment runs on the MI355X for the whole batch (data/__init__.py: DeviceBatches), so DataLoader
workers never touch fp32 and a batch crosses PCIe as bytes (4x less)."""
import random
from io import BytesIO

import torch
from PIL import Image
from torch.utils.data import Dataset

import data.util as Util


class LRHRDataset(Dataset):
    def __init__(self, dataroot, datatype, l_resolution=16, r_resolution=128, split='train', data_len=-1, need_LR=False):
        self.datatype = datatype
        self.l_res = l_resolution
        self.r_res = r_resolution
        self.data_len = data_len
        self.need_LR = need_LR
        self.split = split
        self.env = None
        if datatype == 'lmdb':
            try:
                import lmdb
            except ImportError as e:
                raise ImportError('datatype "lmdb" needs the lmdb module (not installed here); use datatype "img"') from e
            self.env = lmdb.open(dataroot, readonly=True, lock=False, readahead=False, meminit=False)
            with self.env.begin(write=False) as txn:
                self.dataset_len = int(txn.get('length'.encode('utf-8')))
        elif datatype == 'img':
            self.sr_path = Util.get_paths_from_images('{}/sr_{}_{}'.format(dataroot, l_resolution, r_resolution))
            self.hr_path = Util.get_paths_from_images('{}/hr_{}'.format(dataroot, r_resolution))
            if self.need_LR:
                self.lr_path = Util.get_paths_from_images('{}/lr_{}'.format(dataroot, l_resolution))
            self.dataset_len = len(self.hr_path)
        else:
            raise NotImplementedError('data_type [{:s}] is not recognized.'.format(datatype))
        self.data_len = self.dataset_len if self.data_len <= 0 else min(self.data_len, self.dataset_len)

    def __len__(self):
        return self.data_len

    def _lmdb_triplet(self, txn, index):
        key = str(index).zfill(5)
        hr = txn.get('hr_{}_{}'.format(self.r_res, key).encode('utf-8'))
        sr = txn.get('sr_{}_{}_{}'.format(self.l_res, self.r_res, key).encode('utf-8'))
        lr = txn.get('lr_{}_{}'.format(self.l_res, key).encode('utf-8')) if self.need_LR else None
        return hr, sr, lr

    def __getitem__(self, index):
        img_LR = None
        if self.datatype == 'lmdb':
            with self.env.begin(write=False) as txn:
                hr, sr, lr = self._lmdb_triplet(txn, index)
                while hr is None or sr is None:            # skip the invalid index (LRHR_dataset.py:65-80)
                    hr, sr, lr = self._lmdb_triplet(txn, random.randint(0, self.data_len - 1))
            img_HR = Image.open(BytesIO(hr)).convert('RGB')
            img_SR = Image.open(BytesIO(sr)).convert('RGB')
            if self.need_LR:
                img_LR = Image.open(BytesIO(lr)).convert('RGB')
        else:
            img_HR = Image.open(self.hr_path[index]).convert('RGB')
            img_SR = Image.open(self.sr_path[index]).convert('RGB')
            if self.need_LR:
                img_LR = Image.open(self.lr_path[index]).convert('RGB')
        # the ONE RandomHorizontalFlip(p=0.5) draw transform_augment makes per sample in split 'train'
        # (data/util.py:78-81: torch.rand(1) < p on the stacked [SR, HR])
        flip = bool(torch.rand(1).item() < 0.5) if self.split == 'train' else False
        item = {'HR': Util.image_to_u8(img_HR), 'SR': Util.image_to_u8(img_SR), 'Index': index, 'flip': flip}
        if self.need_LR:
            item['LR'] = Util.image_to_u8(img_LR)
            if self.split == 'train' and item['LR'].shape != item['HR'].shape:
                # the reference's torch.stack([LR, SR, HR]) (data/util.py:79) raises for differently sized images
                raise RuntimeError('stack expects each tensor to be equal size (need_LR with split "train")')
        return item
