# coding: utf-8


def getETA(batchTime, nbBatch, batchIndex, nbEpoch, epoch):
    """Remaining training time, 'HhMMmSSs': batches left in this epoch plus all batches of the epochs still to run,
    times the last batch duration (reference utils/tools.py:4-8)."""
    batches_left = (nbBatch - batchIndex) + nbBatch * (nbEpoch - epoch)
    total = int(batchTime * batches_left)
    hours, rest = total // 3600, total % 3600
    return "%dh%02dm%02ds" % (hours, rest // 60, rest % 60)
