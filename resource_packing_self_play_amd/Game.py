"""Plugin base class naming the game API of the bin-packing path (xw_mcts/Game.py:1-113; the BPP subclass dropped the
player arguments, BinPackingGame.py:24-120)."""


class Game:
    def getInitBoard(self): raise NotImplementedError
    def getBoardSize(self): raise NotImplementedError
    def getActionSize(self): raise NotImplementedError
    def getNextState(self, board, action, items_list_board): raise NotImplementedError
    def getValidMoves(self, board): raise NotImplementedError
    def getGameEnded(self, total_board, items_total_area, rewards_list, alpha): raise NotImplementedError
    def stringRepresentation(self, board): raise NotImplementedError
