"""Checkpoint half of the reference's base trainer (generalframework/trainer/trainer.py:208-220):
``best_{i}.pth`` = {'segmentator': Segmentator.state_dict, 'best_score', 'best_epoch'} -- the
format Summary.py:70-74 reloads."""
import os
from pathlib import Path

import torch


class Trainer(object):
    best_score = -1
    segmentator = None
    save_dir = Path('tmp')

    def checkpoint(self, metric, epoch, filename='best.pth'):
        if metric <= self.best_score:
            return
        self.best_score = metric
        state_dict = {'segmentator': self.segmentator.state_dict, 'best_score': metric, 'best_epoch': epoch}
        torch.save(state_dict, Path(os.path.join(self.save_dir, filename)))
