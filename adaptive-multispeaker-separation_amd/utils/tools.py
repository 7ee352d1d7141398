# coding: utf-8


def getETA(batchTime, nbBatch, batchIndex, nbEpoch, epoch):
    """Remaining training time as a string (reference utils/tools.py:4-8)."""
    seconds = int(batchTime * (nbBatch - batchIndex) + batchTime * nbBatch * (nbEpoch - epoch))
    m, s = divmod(seconds, 60)
    h, m = divmod(m, 60)
    return "%dh%02dm%02ds" % (h, m, s)
